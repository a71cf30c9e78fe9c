// upx_pipeline.h - two-thread hand-over of a streamed host call (upx_process / upx_process_chunked /
// upx_process_tracks in upx_lib.hip).
//
// The caller's thread SUBMITS work item i (upload + kernels, asynchronous on the device) while a second thread
// COMPLETES item i-1 (waits for its kernels, downloads its owned range; a download into pageable memory blocks its
// caller, so it cannot share the submitting thread).  Device buffer set i % 2 is reused by item i+2, hence item i
// is only submitted once item i-2 is complete.
//
// Either side may fail.  A failure is recorded once, raises `stop`, and wakes the other side, which returns at its
// next wait: neither thread can wait for a hand-over that will never come (the round-1 code raised its abort flag
// only for submit-side errors, so a download error with two or more items left hung the call).
//
// Plain C++ (no HIP): tests/emu builds the same header with g++ and injects failures (tests/test_pipeline.py).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace upx {

// ---- chunk schedule of the WAV pipeline (upx_wav_shard_open in upx_lib.hip) -------------------------------------------
// One chunk of a shard: owned frames [start, start + own); its kernels read t_in frames from `start` (own range + right
// halo) and write t_out (own + spill, clipped to the planes' end).
struct WavChunkRec {
    int64_t start, own, t_in, t_out;
};
// Chunks of the shard on the shard grid (as items_of_track cuts a streamed host call): the kernels of chunk c run while the
// samples of chunk c + 1 come up.  What the call waits for is  max over the chunks of (its samples have landed, the previous
// chunk's kernels have ended) + its own kernels,  so the cut is a small scheduling problem with two rates: the link (55 GB/s:
// 13.7 M frames per ms of 16-bit stereo, 9.2 M of 24-bit) and the kernels (`kernel_mframes_per_ms`: ~21 for the six-band
// plans) plus a fixed cost per chunk (~0.2 ms: launches that do not fill the chip, decode, peaks, the parked spill - measured
// per launch in the rocprofv3 trace of scripts/wav_overlap_probe.py).  A chunk whose kernels take as long as the NEXT chunk's
// upload keeps both busy; the last chunk is what stays exposed.  Every first-chunk length (in halves of `chunk`) is tried,
// each candidate continued by that balance rule, the two queues simulated, and the schedule that ends first kept.
// `uniform`: chunks of `chunk` frames (tests, A/B).  One chunk when `grid` == 0 (the plan's hops share no grid), `chunk` <= 0
// or the shard is shorter than two chunks.  Chunk starts are multiples of `grid` (= 2 hop_max: frame parity is global), every
// chunk but a single one owns at least max(chunk, 4 spill) frames, no chunk exceeds what one launch can index (2^29 - 1).
inline void wav_schedule(int64_t t_in, int64_t own_len, int64_t t_out, int64_t grid, int64_t spill, int64_t chunk, bool uniform,
                         int bytes_per_frame, double kernel_mframes_per_ms, std::vector<WavChunkRec>& out) {
    out.clear();
    if (chunk > 0 && grid > 0) {
        chunk = (chunk + grid - 1) / grid * grid;
        if (chunk < 4 * spill) chunk = (4 * spill + grid - 1) / grid * grid;
    }
    if (chunk <= 0 || grid <= 0 || own_len < 2 * chunk) {
        out.push_back(WavChunkRec{0, own_len, t_in, t_out});
        return;
    }
    const int64_t launch_cap = ((1LL << 29) - 1 - spill) / grid * grid;      // what one launch can index
    const double r_up = 55e6 / (bytes_per_frame > 0 ? bytes_per_frame : 4);  // frames per ms over the link
    const double r_k = (kernel_mframes_per_ms > 0.5 ? kernel_mframes_per_ms : 21.0) * 1e6, fixed_ms = 0.2;
    auto round_grid = [grid](double v) { return (int64_t)(v / (double)grid + 0.5) * grid; };
    auto build = [&](int64_t first, std::vector<int64_t>& own) {   // the owned lengths of a schedule that starts with `first`
        own.clear();
        int64_t left = own_len, next = first;
        while (left > 0) {
            int64_t c = next < chunk ? chunk : next;
            if (c > launch_cap) c = launch_cap;
            if (left - c < chunk) c = left > launch_cap ? launch_cap : left;   // never a last chunk shorter than the smallest
            own.push_back(c);
            left -= c;
            next = uniform ? chunk : round_grid(r_up * (fixed_ms + (double)c / r_k));   // its upload = this chunk's kernels
        }
    };
    auto simulate = [&](const std::vector<int64_t>& own) {
        double t_up = 0.0, t_k = 0.0;
        for (size_t c = 0; c < own.size(); ++c) {
            t_up += (double)(own[c] + (c == 0 ? spill : 0)) / r_up;
            t_k = (t_up > t_k ? t_up : t_k) + fixed_ms + (double)own[c] / r_k;
        }
        return t_k;
    };
    std::vector<int64_t> best, cand;
    double t_best = 1e300;
    const int64_t step = chunk / 2 / grid * grid > 0 ? chunk / 2 / grid * grid : grid;
    for (int64_t first = chunk; first <= own_len; first += step) {
        build(first, cand);
        const double t = simulate(cand);
        if (t < t_best - 1e-9) {
            t_best = t;
            best = cand;
        }
        if (cand.size() == 1 || uniform) break;
    }
    int64_t start = 0;
    for (size_t c = 0; c < best.size(); ++c) {
        const bool last = c + 1 == best.size();
        WavChunkRec w;
        w.start = start;
        w.own = last ? own_len - start : best[c];
        w.t_in = t_in - start < w.own + spill ? t_in - start : w.own + spill;
        // (the planes end at t_out: a chunk must not write - or park - samples beyond them)
        w.t_out = last || t_out - start < w.own + spill ? t_out - start : w.own + spill;
        out.push_back(w);
        start += w.own;
    }
}

// submit(i, msg) / complete(i, msg): 0 on success, otherwise a upx_status (< 0) with `msg` filled in.
// Returns 0 or the first recorded status; `err` then holds its message.
template <class Submit, class Complete>
int run_pipeline(int64_t n_items, Submit&& submit, Complete&& complete, std::string& err) {
    struct Shared {
        std::mutex m;
        std::condition_variable cv;
        int64_t submitted = 0, completed = 0;
        bool stop = false;
        int rc = 0;
        std::string err;
    } sh;
    auto record = [&](int rc, const std::string& msg) {   // call with sh.m held
        if (sh.rc == 0) {
            sh.rc = rc;
            sh.err = msg;
        }
        sh.stop = true;
        sh.cv.notify_all();
    };
    std::thread completer([&] {
        for (int64_t i = 0; i < n_items; ++i) {
            {
                std::unique_lock<std::mutex> lk(sh.m);
                sh.cv.wait(lk, [&] { return sh.submitted > i || sh.stop; });
                if (sh.stop) return;
            }
            std::string msg;
            const int rc = complete(i, msg);
            std::lock_guard<std::mutex> lk(sh.m);
            if (rc != 0) {
                record(rc, msg);
                return;
            }
            sh.completed = i + 1;
            sh.cv.notify_all();
        }
    });
    for (int64_t i = 0; i < n_items; ++i) {
        if (i >= 2) {   // buffer set i % 2: item i-2 must be complete (its kernels are then done as well)
            std::unique_lock<std::mutex> lk(sh.m);
            sh.cv.wait(lk, [&] { return sh.completed >= i - 1 || sh.stop; });
            if (sh.stop) break;
        }
        std::string msg;
        const int rc = submit(i, msg);
        std::lock_guard<std::mutex> lk(sh.m);
        if (rc != 0) {
            record(rc, msg);
            break;
        }
        if (sh.stop) break;
        sh.submitted = i + 1;
        sh.cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(sh.m);
        if (sh.submitted < n_items) {   // nothing more will be handed over
            sh.stop = true;
            sh.cv.notify_all();
        }
    }
    completer.join();
    err = sh.err;
    return sh.rc;
}

}   // namespace upx
