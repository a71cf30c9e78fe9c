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

namespace upx {

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
