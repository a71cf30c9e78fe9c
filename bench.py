#!/usr/bin/env python3
"""
bench.py - headline benchmark: stereo Msamples/s upmixed (6 bands, STFT <= 8192).

    python bench.py --gpus N --steps K --warmup W [--workload c1|c2|c3|default|c4share|batch|ov50|ov875|ov60|wide65536]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under a launcher that exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run,
upmix_amd.launch, srun ...) this process is one rank; WITHOUT one (the bare `python bench.py --gpus N`) it is only a
parent that starts the N ranks itself through upmix_amd.launch.run - fresh child processes, never a re-exec, and the
parent makes no GPU call - and exits with their code; rank 0 prints the JSON line on the stdout they share.

A "step" is one pass of the hot path over synthetic stereo already resident in HBM.  Workloads:

  c1       BASELINE configs[0] (the reference's own CPU-runnable case): 10 s of 48 kHz stereo, ONE band 0-24 kHz,
           STFT 2048, 75 % overlap; its cpu_baseline runs the WHOLE config on the host.
  c2       BASELINE configs[1]: 60 s of 48 kHz stereo, 3 bands (crossovers 300 / 3000 Hz), threshold_factor 64,
           max STFT 4096 -> STFT [4096, 4096, 1024].
  c3       (default; the line the driver records) BASELINE.json configs[2]: 10 min of 48 kHz stereo, 6 bands, edges
           0/30/120/480/1920/7680 Hz, STFT [8192,8192,8192,4096,1024,256], Blackman-Harris, 75 % overlap, raised-cosine
           crossovers.  N > 1 ranks: N x 10 min, time-sharded on the hop_max grid (one shard per GPU, weak scaling),
           every step ends with the single RCCL all-reduce of the overlap-add seam.
  default  the same signal through the reference's own default plan (center_extraction.py:555, main.py:62:
           STFT [65536,65536,16384,4096,1024,256]) - what a caller who does not cap the STFT size runs.
  c4share  one GPU's share of configs[3]: 15 min of 96 kHz stereo (86.4 M samples), STFT [8192 x4, 2048, 512];
           N > 1 ranks: the time-sharded 2 h (at N = 8) signal with the RCCL seam.
  batch    configs[4]: 8 independent 5-min 48 kHz tracks per GPU (64 on 8 GPUs), C3 plan, replicas only (no
           communication); `value` = tracks resident in HBM, the PCIe-inclusive upx_process_tracks rate is in `e2e`.
  ov50 / ov875 / ov60 / wide65536   arguments the reference accepts outside the BASELINE shapes (any overlap,
           center_extraction.py:252; any max_block_size, :173): C3's signal and edges at overlap 0.5 / 0.875 / 0.6, and
           chain_bands([0, 3000]) at the default max_block_size - the perf table of DESIGN.md 7, not BASELINE lines.

The ranks of an N > 1 run meet over upmix_amd.rendezvous (standard-library sockets on the launcher's MASTER_ADDR /
MASTER_PORT: the 128-byte RCCL id, the barriers and the max-over-ranks of the wall time); this process never imports
torch, all GPU work goes through libupmix_hip.so.  Prints ONE JSON line on rank 0 (fields: README / DESIGN.md section 7).
"""
import argparse
import hashlib
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EDGES = [0, 30, 120, 480, 1920, 7680]
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
VALU_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: fp32 vector peak (SURVEY.md 8(d) secondary ceiling)
TIMING_STRIDE = 4                # HIP events around the launches of every 4th step of the timed region
ALGO_BYTES_IN, ALGO_BYTES_OUT = 8, 12   # per band 8 B stereo in + 12 B Ls/C/Rs out = 20 B per sample
KERNEL_SOURCES = ["upmix_amd/csrc/upx_core.h", "upmix_amd/csrc/upx_zoom.h", "upmix_amd/csrc/upx_big.h",
                  "upmix_amd/csrc/upx_kernels.h", "upmix_amd/csrc/upx_lib.hip", "upmix_amd/csrc/upx_reg_big.hip",
                  "upmix_amd/csrc/upx_reg_fused.hip", "upmix_amd/csrc/upx_reg_fused_single.hip",
                  "upmix_amd/csrc/upx_reg_zoom256.hip", "upmix_amd/csrc/upx_reg_zoom512.hip",
                  "upmix_amd/csrc/upx_reg_zoom1024.hip"]

WORKLOADS = {
    #          sr     seconds  max_stft  BASELINE config                                       band edges (None: the six-band EDGES), threshold_factor, seed
    "c1":      (48000, 10,     2048,     "BASELINE configs[0]",                                 "single", 32, 0),
    "c2":      (48000, 60,     4096,     "BASELINE configs[1]",                                 [0, 300, 3000], 64, 1),
    "c3":      (48000, 600,    8192,     "BASELINE configs[2]",                                 None, 32, 2),
    "default": (48000, 600,    65536,    "configs[2] signal, reference default plan (max STFT 65536)", None, 32, 2),
    "c4share": (96000, 900,    8192,     "one GPU's share of BASELINE configs[3] (2 h at 96 kHz over 8 GPUs)", None, 32, 2),
    "batch":   (48000, 300,    8192,     "BASELINE configs[4] (8 tracks of 5 min per GPU)",    None, 32, 2),
    # the argument space the reference accepts beyond the BASELINE shapes (center_extraction.py:56-75, :252, :173-197): C3's
    # signal and edges at other overlaps, and a wide band at the reference's default max_block_size
    "ov50":    (48000, 600,    8192,     "configs[2] signal and edges at overlap 0.5 (hop N/2)", None, 32, 2),
    "ov875":   (48000, 600,    8192,     "configs[2] signal and edges at overlap 0.875 (hop N/8)", None, 32, 2),
    "ov60":    (48000, 60,     8192,     "60 s of configs[2]'s signal and edges at overlap 0.6 (hop int(0.4 N): the unfused run-time-hop path)", None, 32, 2),
    "wide65536": (48000, 600,  65536,    "configs[2] signal, chain_bands([0, 3000]) with the reference's default max_block_size: "
                                         "a 0-3 kHz band at STFT 65536 (pass band too wide for the band-limited path)", [0, 3000], 32, 2),
}
OVERLAP = {"ov50": 0.5, "ov875": 0.875, "ov60": 0.6}      # every other workload: 0.75
TRACKS_PER_GPU = 8


def workload_bands(workload, make_single, chain):
    """
    The band list of a workload through the caller's constructors (upmix_amd's or the oracle's - the same call shapes):
    `make_single(block, overlap, f_lo, f_hi, sr, xover_mode, width_lo, width_hi)` for configs[0]'s one full band
    (center_extraction.py:217-271 called directly, as SURVEY.md 8 C1 states it: STFT 2048, 0-24 kHz, no fade at 0 Hz,
    none reached at Nyquist), `chain(edges, sr, max_stft, threshold_factor)` = chain_bands (:518-580) for the others.
    """
    sr, _seconds, max_stft, _cfg, edges, factor, _seed = WORKLOADS[workload]
    if edges == "single":
        return [make_single(max_stft, 0.75, 0.0, sr / 2.0, sr, "raised_cosine", 0.0, 0.25 * sr / 2.0)]
    return chain(EDGES if edges is None else edges, sr, max_stft, factor)


def host_description():
    """What the CPU line ran on: logical CPUs visible, the model string of /proc/cpuinfo, NumPy's version."""
    model = None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count()
    return {"host_cpus": os.cpu_count(), "usable_cpus": usable, "cpu_model": model, "numpy": np.__version__}


def synth(total, seed):
    rng = np.random.default_rng(seed)
    m = rng.standard_normal(total)
    s = rng.standard_normal(total)
    x = np.empty((total, 2), dtype=np.float32)
    x[:, 0] = 0.1 * (m + 0.5 * s)
    x[:, 1] = 0.1 * (m - 0.5 * s)
    return x


def kernel_sources_sha():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_baseline(workload, target_seconds=15.0):
    """
    The oracle in the reference's scheduling shape (ThreadPoolExecutor(), one task per band, sequential
    frame loop per band: center_extraction.py:499-501 -> :449-460) on a bounded sample of the same workload: a
    calibration slice (10 s of audio, or the whole workload if that is shorter) sizes the timed sample so that it costs
    about `target_seconds` of CPU wall time, capped at the workload's own length (configs[0] runs whole).
    """
    from oracle import upmix_oracle as orc
    sr, seconds, _max_stft, _cfg, _edges, _factor, seed = WORKLOADS[workload]
    bands = workload_bands(workload, orc.Band,
                           lambda e, sr_, m, f: orc.plan_bands(e, OVERLAP.get(workload, 0.75), orc.win_blackman_harris, sr_, max_block_size=m,
                                                               threshold_factor=f))

    def run(secs):
        total = int(sr * secs)
        x = synth(total, seed).astype(np.float64)
        t0 = time.perf_counter()
        orc.extract_multi_band_threadpool(x[:, 0], x[:, 1], bands)
        return total, time.perf_counter() - t0

    calib = float(min(10.0, seconds))
    total, dt = run(calib)
    sample_seconds = float(min(seconds, max(calib, calib * target_seconds / max(dt, 1e-3))))
    if sample_seconds > calib:
        total, dt = run(sample_seconds)
    else:
        sample_seconds = calib
    # the same frame loops one band after the other (no thread pool): SURVEY 8(d) asks for both
    xs = synth(int(sr * calib), seed).astype(np.float64)
    t0 = time.perf_counter()
    orc.extract_multi_band(xs[:, 0], xs[:, 1], bands, per_band=orc.band_process_streaming)
    serial = len(xs) / (time.perf_counter() - t0) / 1e6
    # ThreadPoolExecutor() defaults to min(32, cpus + 4) workers (the reference's own call); one task per band
    host = host_description()
    threads = min(len(bands), min(32, (os.cpu_count() or 1) + 4))
    whole = sample_seconds >= seconds
    out = {
        "value": round(total / dt / 1e6, 4), "unit": "Msamples/s",
        "cores": host["usable_cpus"], "threads_used": threads,
        "cores_note": "cores = the CPUs of the box this process may run on (north_star: the core count of the host the CPU "
                      "line was timed on); threads_used = one task per band in the reference's ThreadPoolExecutor() "
                      "(center_extraction.py:499-501), most of whose work the GIL serialises",
        "kind": "port",
        "bands_serial_value": round(serial, 4),
        "sample": (f"the WHOLE workload ({sample_seconds:g} s of audio, seed {seed})" if whole else
                   f"first {sample_seconds:g} s of the same workload (seed {seed})") +
                  f", oracle/upmix_oracle.py extract_multi_band_threadpool: ThreadPoolExecutor(), one task per band "
                  f"(= {threads} threads), float64 numpy.fft, {dt:.1f} s wall",
    }
    out.update(host)
    return out


def executed_flops_per_sample(kernel_name, n, k, n_bands_in_launch, phase):
    """
    Flops of the transforms a launch actually runs, per stereo sample (model, same 5 n log2 n per complex transform
    as SURVEY.md 8(d)): a fused launch runs 2.5 complex N-point transforms per frame and K frames cover a sample ->
    12.5 K log2 N (+ 80 for the mask, per band carried); the band-limited pair runs D transforms of P points where a
    full-size path runs one of N = D P points -> 5 K log2 P for the analysis (+ 8 K for ramp and residue sum, + the
    mask over P/2 of N/2 bins per band), 7.5 K log2 P (+ 9 K for the ramp) for the synthesis.  `phase`: 0 analysis,
    1 synthesis / single kernel.  Unfused launches (upx_big_*) run full-size transforms.
    """
    m = re.search(r"ZoomCfg<(\d+), *(\d+), *(\d+)>", kernel_name)
    if m:
        p = 1 << int(m.group(1))
        if phase == 0:
            return 5.0 * k * np.log2(p) + 8.0 * k + 80.0 * n_bands_in_launch * p / n
        return 7.5 * k * np.log2(p) + 9.0 * k
    return 12.5 * k * np.log2(n) + 80.0 * n_bands_in_launch


def canonical_kernel_name(name):
    """One spelling for a kernel whether the library's table or rocprofv3 names it: no blanks ("> >"), no default
    template arguments (rocprofv3 prints upx::Live<0, 1048576> for the general flavour), no namespace of the symbol."""
    name = name.replace(" ", "").replace(",upx::Live<0,1048576>", "")
    name = re.sub(r"^(upxk::)?(upx_band_kernel<.*),true>$", r"\1\2>", name)     # MERGED = true is the template's default too
    return re.sub(r"^(void)?(\(anonymousnamespace\)::|upxk::)?", "", name)


def load_pmc_traffic(kernel_tag, workload="c3"):
    """
    HBM bytes per launch of `kernel_tag` in `workload` from the committed rocprofv3 PMC summary
    (profiles/pmc_traffic.json: {workload: {kernel: ...}}) - only if that summary was collected from the kernel sources
    that are running now (their hash is stored with it).
    """
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    sha = rec.get("_kernel_sources_sha256_16")
    if sha != kernel_sources_sha():
        return None, f"profiles/pmc_traffic.json was collected from other kernel sources ({sha}); re-run scripts/pmc.sh"
    want = canonical_kernel_name(kernel_tag)
    per_workload = rec.get(workload)
    if not isinstance(per_workload, dict):
        return None, f"no PMC passes of workload {workload!r} in profiles/pmc_traffic.json"
    v = next((e.get("hbm_bytes_per_launch") for k, e in per_workload.items() if canonical_kernel_name(k) == want), None)
    return v, (None if v is not None else "kernel not in profiles/pmc_traffic.json")


def e2e_rates(ux, plan, bands, sr, nominal, seed=2, fresh_process=True):
    """PCIe-inclusive rates of the host-buffer entry points on this workload's signal (never `value`)."""
    from upmix_amd import wav as _wav  # noqa: F401
    out = {}
    x = synth(nominal, seed)

    def timed(fn, reps=3):
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return best

    # upx_process (streamed in chunks: upload / kernels / download overlap) into arrays that were touched before ...
    lib, check = plan._lib, __import__("upmix_amd")._lib.check
    f32p = __import__("upmix_amd")._lib.f32p
    warm = [np.zeros(nominal, dtype=np.float32) for _ in range(3)]
    run = lambda outs: check(lib.upx_process(plan.handle, x.ctypes.data_as(f32p), nominal, *(o.ctypes.data_as(f32p) for o in outs)))  # noqa: E731
    run(warm)
    dt = timed(lambda: run(warm))
    out["upx_process_warm_buffers"] = {"ms": round(dt * 1e3, 2), "Msamples_per_s": round(nominal / dt / 1e6, 1)}
    # ... into fresh pageable NumPy arrays (the kernel has to fault in and zero the new pages, and unmap them after) ...
    dt = timed(lambda: run([np.empty(nominal, dtype=np.float32) for _ in range(3)]))
    out["upx_process_fresh_pageable_arrays"] = {"ms": round(dt * 1e3, 2), "Msamples_per_s": round(nominal / dt / 1e6, 1)}
    # ... and what the drop-in entry does: fresh result arrays from the pool of page-locked blocks (upmix_amd/hostmem.py;
    # the first call pins its blocks, later calls reuse the blocks of results that have been dropped)
    t0 = time.perf_counter()
    res = plan.process(x)
    first = time.perf_counter() - t0
    del res
    dt = timed(lambda: plan.process(x))
    out["upx_process_fresh_arrays"] = {"ms": round(dt * 1e3, 2), "Msamples_per_s": round(nominal / dt / 1e6, 1),
                                       "first_call_ms": round(first * 1e3, 2),
                                       "note": "DevicePlan.process: new result arrays per call, in pooled page-locked memory"}
    # The documented drop-in call (INTEGRATION.md option A): main.py:49-50, 78-80 hands the entry two float64 COLUMN VIEWS
    # of one [T, 2] array.  Steady cost here (plan cached, pool warm); what a fresh process pays for its one call - import,
    # chain_bands, plan creation, the call itself into pageable arrays - is measured in a child process below.
    wave64 = x.astype(np.float64)
    entry = lambda: ux.extract_center_left_right_multi_band_in_memory(wave64[:, 0], wave64[:, 1], sr, bands)  # noqa: E731
    t0 = time.perf_counter()
    res = entry()
    first = time.perf_counter() - t0
    del res
    entry()
    dt = timed(entry)
    link_ms = wave64.nbytes / 2 / 55e9 * 1e3      # the extra bytes float64 input puts on the link, at its ~55 GB/s
    out["drop_in_entry_float64_views"] = {
        "ms": round(dt * 1e3, 2), "Msamples_per_s": round(nominal / dt / 1e6, 1),
        "first_call_in_this_process_ms": round(first * 1e3, 2),
        "extra_upload_ms_at_link_rate": round(link_ms, 2),
        "vs_upx_process_fresh_arrays": round(dt * 1e3 / out["upx_process_fresh_arrays"]["ms"], 2),
        "note": "extract_center_left_right_multi_band_in_memory(wave[:, 0], wave[:, 1], sr, bands) on a float64 [T, 2] parent: "
                "the columns go up as they are (upx_process_lr), cast + interleave on the device; first call in this process = "
                "plan creation for the cached plan + result arrays in pageable memory (lazy pinning)"}
    del wave64
    # what the drop-in entry costs in Python around the library call: the same entry on an empty signal (plan-cache key
    # from the memoised band signatures + the UPX_* environment, check-out of the cached plan, argument inspection,
    # three empty result arrays; no library call for T = 0).  Round 5 re-hashed every band's windows per call here:
    # 0.17 ms (C3 plan) / 0.70 ms (default plan).
    empty64 = np.zeros((0, 2), dtype=np.float64)
    entry0 = lambda: ux.extract_center_left_right_multi_band_in_memory(empty64[:, 0], empty64[:, 1], sr, bands)  # noqa: E731
    entry0()
    t0 = time.perf_counter()
    for _ in range(2000):
        entry0()
    out["drop_in_entry_python_overhead_us"] = round((time.perf_counter() - t0) / 2000 * 1e6, 2)
    if fresh_process:
      try:
        import subprocess
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "drop_in_fresh_process.py"), "--seconds",
                            str(nominal / sr), "--sr", str(sr), "--max-stft", str(max(b.block_size for b in bands))],
                           capture_output=True, text=True, timeout=300)
        out["drop_in_entry_float64_views"]["fresh_process"] = json.loads(r.stdout.strip().splitlines()[-1])
      except Exception as exc:
        out["drop_in_entry_float64_views"]["fresh_process"] = {"error": repr(exc)}
    # WAV pipeline: PCM16 in, decode + all bands + peak scale + stereo_sum layout + quantisation on the device, PCM16 out
    pcm = np.clip(np.rint(x * 32767.0), -32768, 32767).astype("<i2")
    del x
    # (pageable source: every chunk's upload is a blocking staged copy on the calling thread, in turn with its launches)
    fn = lambda: plan.wav_pipeline(pcm, 16, 2, nominal, "stereo_sum", 16)  # noqa: E731
    fn()
    dt = timed(fn)
    out["upx_wav_pipeline_pcm16_stereo_sum_pageable_input"] = {"ms": round(dt * 1e3, 2),
                                                              "Msamples_per_s": round(nominal / dt / 1e6, 1)}
    # ... and what the file entries do (cli, batch, multi_gpu read the file's bytes into page-locked memory): asynchronous
    # uploads, chunk c's kernels under the upload of chunk c + 1
    pinned = plan.host_empty(pcm.nbytes).view("<i2").reshape(pcm.shape)
    pinned[...] = pcm
    fn = lambda: plan.wav_pipeline(pinned, 16, 2, nominal, "stereo_sum", 16)  # noqa: E731
    fn()
    dt = timed(fn)
    t = plan.wav_pipeline_times_ms()
    out["upx_wav_pipeline_pcm16_stereo_sum"] = {"ms": round(dt * 1e3, 2), "Msamples_per_s": round(nominal / dt / 1e6, 1),
                                                "begin_ms": round(t["begin"], 2), "begin_tail_ms": round(t["begin_tail"], 2),
                                                "finish_ms": round(t["finish"], 2),
                                                "note": "begin = upload || decode || bands || peaks, chunk by chunk; begin_tail = "
                                                        "what of it came after the last sample landed; finish = export || download"}
    out["note"] = ("pageable input buffers; best of 3; PCIe-inclusive, reported beside `value` (which is HBM-resident), "
                   "SURVEY.md 8(d)")
    return out


def e2e_multi_gpu_file(bands, sr, nominal, device):
    """
    The product's one-WAV-over-N-GPUs entry (upmix_amd.multi_gpu.run_rank) on this GPU's share, file to file: a PCM_24
    WAV of the workload's signal is written to a scratch directory, then one rank reads its bytes into page-locked
    memory, runs the sharded device pipeline (upx_wav_shard_begin / _finish) and writes its slice of the PCM_24 output.
    Second of two runs (page cache and the pinned pool warm); seconds per phase as run_rank measures them.
    """
    import shutil
    import tempfile
    from upmix_amd import multi_gpu, wav
    tmp = tempfile.mkdtemp(prefix="upx_bench_", dir=os.environ.get("UPX_BENCH_TMP", None))
    try:
        path = os.path.join(tmp, "share.wav")
        off = wav.create(path, nominal, sr, "PCM_24", 2)
        step = 1 << 22
        for a in range(0, nominal, step):                       # in pieces: no full-length float64 temporaries
            n = min(step, nominal - a)
            wav.write_at(path, off + a * 6, wav.encode(synth(n, (2, a)).astype(np.float64), "PCM_24")[2])
        best = None
        for _ in range(2):
            times = {}
            t0 = time.perf_counter()
            multi_gpu.run_rank(path, os.path.join(tmp, "out"), "stereo_sum", bands, 0.75, "PCM_24", 0, 1, None,
                               device=device, log=lambda *_: None, times=times)
            times["total_s"] = time.perf_counter() - t0
            best = times
        return {"multi_gpu_run_rank_pcm24_stereo_sum": {k: round(v * 1e3, 1) for k, v in best.items()},
                "unit": "ms", "Msamples_per_s": round(nominal / best["total_s"] / 1e6, 1),
                "note": "file -> file on one rank: read_s = the shard's bytes from the (cached) file into page-locked memory; "
                        "device_begin_s = upload + decode + all bands + peaks; device_finish_s = scale + export + "
                        "quantise + download; write_s = header + this rank's slice; plan creation included in total_s"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--seconds", type=float, default=None, help="audio per GPU / per track (default: the workload's)")
    ap.add_argument("--preheat-ms", type=float, default=200.0,
                    help="untimed steps for this long before the W warm-up steps: the card leaves its idle power state")
    ap.add_argument("--no-reserve", action="store_true",
                    help="skip upx_plan_reserve (its tiny warm-up call launches every kernel once on a short signal, which would "
                         "dilute per-kernel averages of a profiler run that only sees a few launches: scripts/pmc.sh, prof.sh)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive side measurements")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # The bare command: nobody started the ranks, so this process does - N fresh children of itself with RANK /
        # LOCAL_RANK / WORLD_SIZE / MASTER_* set (upmix_amd.launch.run: no re-exec, and nothing in this parent has
        # touched the GPU runtime), rank 0 among them prints the JSON line on the shared stdout.
        from upmix_amd import launch
        sys.stdout.flush()
        return launch.run(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a launcher's world size wins over the flag (the driver passes both, equal)
        print(f"[bench] WARNING: --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr)
        args.gpus = world

    from upmix_amd.rendezvous import Rendezvous
    group = Rendezvous.from_env()    # rank 0 listens, the others connect; world == 1: no socket

    import upmix_amd as ux
    from upmix_amd import sharding, _lib
    n_dev = _lib.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if local_rank >= n_dev:
        # rehearsal of the N-rank path on a box with fewer GPUs (ranks share a device); not a valid bench line
        print(f"[bench] WARNING: LOCAL_RANK {local_rank} >= {n_dev} visible device(s): sharing device {local_rank % n_dev}",
              file=sys.stderr)
    local_rank = local_rank % n_dev

    sr, seconds, max_stft, cfg_name, wl_edges, wl_factor, seed = WORKLOADS[args.workload]
    if args.seconds is not None:
        seconds = args.seconds
    nominal = int(sr * seconds)
    batch = args.workload == "batch"
    bands = workload_bands(
        args.workload,
        lambda n, ov, lo, hi, sr_, mode, wlo, whi: ux.MultiBandExtractorAccu(n, ov, ux.make_blackman_harris, lo, hi, sr_,
                                                                             mode, wlo, whi, device=local_rank),
        lambda e, sr_, m, f: ux.chain_bands(e, OVERLAP.get(args.workload, 0.75), ux.make_blackman_harris, sr_, max_block_size=m, threshold_factor=f,
                                            verbose=False, device=local_rank))
    plan = ux.DevicePlan(bands, device=local_rank)
    n_bands = len(bands)

    comm = None
    if batch:
        # replicas only: rank r owns tracks r, r + world, ... of TRACKS_PER_GPU x world tracks (seed (4, track))
        from upmix_amd import batch as _batch
        mine = _batch.assign_tracks(TRACKS_PER_GPU * world, rank, world)
        own = t_in = t_out = nominal
        d_tracks = []
        for t in mine:
            d = plan.alloc(nominal * 8)
            plan.h2d(d, synth(nominal, (4, t)))
            d_tracks.append(d)
        d_out = [plan.alloc(nominal * 4) for _ in range(3)]
        samples_per_step_rank = nominal * len(mine)

        def step():
            for d in d_tracks:
                plan.process_device(d, nominal, nominal, d_out[0], d_out[1], d_out[2], nominal)
        calls_per_step = len(d_tracks)
    else:
        if world == 1:
            # one rank: the whole signal, no shard grid needed (hops that share none - overlap 0.6 - run here as well)
            shard = sharding.Shard(0, 0, nominal, nominal, nominal, True)
            shards, spill = [shard], 0
        else:
            geo = sharding.ShardGeometry([b.block_size for b in bands], [b.hop_size for b in bands])
            # N ranks: one signal of N x `seconds` cut on the shard grid; rank g owns shard g (+ right halo, + spill)
            shards = geo.plan(nominal * world, world)
            shard = shards[rank]
            spill = geo.spill
        own, t_in, t_out = shard.own_len, shard.t_in, shard.t_out
        # synthetic stereo: shard g = seed (2, g) (N=1: seed 2); right halo = head of the next shard
        x = synth(own, seed if world == 1 else (seed, rank))
        if t_in > own:
            x = np.concatenate([x, synth(shards[rank + 1].own_len, (seed, rank + 1))[:t_in - own]])
        d_in = plan.alloc(t_in * 8)
        d_out = [plan.alloc((own + spill) * 4) for _ in range(3)]
        plan.h2d(d_in, x)
        del x
        samples_per_step_rank = own
        if world > 1 and os.environ.get("UPX_BENCH_REHEARSAL") == "1":
            # Rehearsal on a box with fewer GPUs than ranks (ranks share a device, which RCCL refuses): the seam goes
            # through host memory + gloo.  Exercises everything but RCCL; the JSON line says so and is not a result.
            class _HostSeam:
                def exchange(self, planes, own_len, spill_):
                    host = [np.empty(own_len + spill_, dtype=np.float32) for _ in range(3)]
                    for h, d in zip(host, planes):
                        plan.d2h(h, d)
                    mine = sharding.pack_seam(host, shard, world, spill_)
                    rows = [np.frombuffer(b, dtype=np.float32).reshape(mine.shape) for b in group.allgather_bytes(mine.tobytes())]
                    sharding.apply_seam(host, shard, np.sum(rows, axis=0, dtype=np.float32))
                    for h, d in zip(host, planes):
                        plan.h2d(d, h)

                def close(self):
                    pass
            comm = _HostSeam()
        elif world > 1:
            comm = sharding.RcclSeam(plan, rank, world, broadcast=group.broadcast_bytes, all_ok=group.all_ok)
            _lib.check(_lib.load().upx_comm_reserve(comm.handle, spill))   # the seam buffer up front, not inside the first step

        def step():
            plan.process_device(d_in, t_in, own, d_out[0], d_out[1], d_out[2], t_out)
            if comm is not None:
                comm.exchange(d_out, own, spill)
        calls_per_step = 1

    # the first call of a shape otherwise allocates its seam / scratch buffers and uploads its stream tables on the way
    # (synchronising calls): prepared up front, so that what the first steps cost beyond a steady step is the card's clocks
    if not args.no_reserve:
        plan.reserve(*((nominal, nominal, nominal) if batch else (t_in, own, t_out)))

    def barrier():
        if hasattr(comm, "wait"):
            comm.wait()      # bounded: a peer that never entered the all-reduce aborts the communicator instead of hanging the sync
        plan.sync()
        group.barrier()

    # An idle card needs ~20 steps (30 ms of work) to reach its running clocks (scripts/clock_ramp_check.py: 1.85,
    # 1.63, 1.54, 1.51 ms for the first groups of five steps, 1.49 from the sixth on; the same again after 2 s of
    # idling).  The metric is sustained throughput, so the ramp is run down before the W warm-up steps, untimed, and
    # reported in the line (`preheat`); --preheat-ms 0 switches it off.
    preheat_steps = 0
    cold_ms = None
    if args.preheat_ms > 0:
        t_pre = time.perf_counter()
        for _ in range(5):
            step()
        plan.sync()
        # every rank runs the same number of steps (a step of the sharded path ends in a collective)
        per_step_ms = group.allreduce_max([(time.perf_counter() - t_pre) * 1e3 / 5])[0]
        cold_ms = per_step_ms      # what a caller's first calls on an idle card cost (clock ramp, first-touch of the tables)
        preheat_steps = max(5, min(2000, int(np.ceil(args.preheat_ms / max(per_step_ms, 1e-3)))))
        for _ in range(preheat_steps - 5):
            step()
        plan.sync()
    for _ in range(args.warmup):
        step()
    plan.enable_timing(True)
    barrier()
    t0 = time.perf_counter()
    # HIP events on the plan's stream around every kernel launch group, kept per call by the library (64 calls) and
    # read after the loop: no synchronisation, and no reads, inside the timed region
    # ... and only every TIMING_STRIDE-th step records them: the ten events of a C3 step cost 24 us of its 1.45 ms
    # (scripts/timing_cost_check.py), which a production call does not pay
    timed_steps = 0
    for i in range(args.steps):
        sampled = i % TIMING_STRIDE == 0
        plan.pause_timing(not sampled)
        timed_steps += sampled
        step()
    if hasattr(comm, "wait"):
        comm.wait()
    plan.sync()
    group.barrier()
    elapsed = group.allreduce_max([time.perf_counter() - t0])[0]    # the slowest rank's time
    n_calls = min(64, timed_steps * calls_per_step)         # the most recent timed process_device calls
    per_call = plan.band_times_calls_ms(n_calls) if n_calls else np.zeros((0, n_bands), np.float32)
    band_ms = per_call.mean(axis=0) if n_calls else np.zeros(n_bands)
    ana_sum, syn_sum = plan.band_phase_times_sum_ms(n_calls) if n_calls else (np.zeros(n_bands), np.zeros(n_bands))
    # median over steps of the step's kernel time (3 warm-ups + median of 10): the last <= 10 steps
    step_ms = per_call.sum(axis=1)
    if calls_per_step > 1 and len(step_ms) >= calls_per_step:
        step_ms = step_ms[len(step_ms) % calls_per_step:].reshape(-1, calls_per_step).sum(axis=1)
    median_ms = float(np.median(step_ms[-10:])) if len(step_ms) else None

    # How much of a cold call is the card and how much is software: the same five steps again after 2 s of idling, with
    # every buffer, table and code object touched - what is left of the difference to a steady step is the clock ramp alone.
    idle_ms = None
    if args.preheat_ms > 0 and world == 1:
        time.sleep(2.0)
        t_idle = time.perf_counter()
        for _ in range(5):
            step()
        plan.sync()
        idle_ms = (time.perf_counter() - t_idle) * 1e3 / 5
    # host work of one process_device call (launch geometry, stream tables: std::vector work per call): the same code path
    # without the launches (upx_plan_reserve's dry run + one synchronisation of the idle stream)
    prep_us = None
    if world == 1 and not args.no_reserve:
        shape = (nominal, nominal, nominal) if batch else (t_in, own, t_out)
        plan.sync()
        t_h = time.perf_counter()
        for _ in range(50):
            plan.reserve(*shape)
        prep_us = (time.perf_counter() - t_h) / 50 * 1e6
    # what a pass with these kernels' lane pattern (4 bytes per lane, grid-stride) reaches on this card, measured in this
    # run: x *= 1.0f over one output plane (upx_scale: one load and one store per sample, values unchanged)
    stream_gbps = None
    if world == 1 and not batch and not args.no_reserve:
        n_probe = int(own)
        for _ in range(3):
            plan.scale(d_out[0], n_probe, 1.0)
        plan.sync()
        t_p = time.perf_counter()
        for _ in range(20):
            plan.scale(d_out[0], n_probe, 1.0)
        plan.sync()
        stream_gbps = 8.0 * n_probe * 20 / (time.perf_counter() - t_p) / 1e9

    if rank == 0:
        total_samples = samples_per_step_rank * world if not batch else nominal * TRACKS_PER_GPU * world
        ms_per_step = elapsed / args.steps * 1e3
        value = total_samples * args.steps / elapsed / 1e6
        sizes = [b.block_size for b in bands]
        groups = {}
        for i in range(n_bands):
            leader, size = plan.band_group(i)
            groups[leader] = size
        # one entry per KERNEL: single-kernel launches carry the band's 20 B per sample, the two kernels of a
        # band-limited launch its 8 B (analysis reads the input) and 12 B (synthesis writes Ls/C/Rs); merged bands
        # multiply.  `ms` = average launch time per process_device call (HIP events, same stream as the kernels).
        launches = []
        hops = [b.hop_size for b in bands]
        for g, n in sorted(groups.items()):
            label = f"bands {g}..{g + n - 1} (STFT {sizes[g]})"
            k_ov = -(-sizes[g] // hops[g])
            a_name = plan.band_phase_kernel_name(g, 0)
            fill = plan.band_fill(g)
            if a_name:
                s_name = plan.band_phase_kernel_name(g, 1)
                launches.append({"kernel": a_name, "bands": label, "ms": float(ana_sum[g]) / n_calls,
                                 "workgroups": fill["workgroups_analysis"], "slots": fill["slots_analysis"],
                                 "algo_bytes": ALGO_BYTES_IN * own * n,
                                 "flops_executed": executed_flops_per_sample(a_name, sizes[g], k_ov, n, 0) * own})
                launches.append({"kernel": s_name, "bands": label,
                                 "workgroups": fill["workgroups"], "slots": fill["slots"],
                                 "ms": float(syn_sum[g]) / n_calls, "algo_bytes": ALGO_BYTES_OUT * own * n,
                                 "flops_executed": executed_flops_per_sample(s_name, sizes[g], k_ov, n, 1) * own})
            else:
                name = plan.band_kernel_name(g)
                launches.append({"kernel": name, "bands": label, "ms": float(band_ms[g]),
                                 "workgroups": fill["workgroups"], "slots": fill["slots"],
                                 "algo_bytes": (ALGO_BYTES_IN + ALGO_BYTES_OUT) * own * n,
                                 "flops_executed": executed_flops_per_sample(name, sizes[g], k_ov, n, 1) * own})
        for L in launches:
            # how much of the chip the launch fills: workgroups dispatched / workgroup slots resident at once
            L["fill"] = round(L["workgroups"] / L["slots"], 3) if L.get("slots") else None
            L["GBps"] = round(L["algo_bytes"] / (L["ms"] * 1e-3) / 1e9, 1) if L["ms"] > 0 else None
            L["frac"] = round(L["GBps"] / HBM_PEAK_GBPS, 4) if L["GBps"] else None
            # executed arithmetic of this launch against the f32 vector peak, and its HBM traffic from the PMC counters
            L["flops_executed"] = float(round(L["flops_executed"]))
            L["TFLOPs"] = round(L["flops_executed"] / (L["ms"] * 1e-3) / 1e12, 2) if L["ms"] > 0 else None
            L["valu_frac"] = round(L["TFLOPs"] / VALU_PEAK_TFLOPS, 4) if L["TFLOPs"] else None
            L["traffic"], note = load_pmc_traffic(L["kernel"], args.workload)
            if L["traffic"] and sum(M["kernel"] == L["kernel"] for M in launches) > 1:
                # the counters are averaged per kernel NAME: a kernel that serves several launch groups of one step
                # (the default plan's three band-limited groups) has no per-group figure
                L["traffic"], note = None, "kernel serves several launch groups of this step: PMC average not attributable"
            L["traffic_ratio"] = round(L["traffic"] / L["algo_bytes"], 3) if L["traffic"] else None
            if note:
                L["traffic_note"] = note
            L["ms"] = round(L["ms"], 4)
        dom = max(launches, key=lambda L: L["ms"])
        traffic, traffic_note = dom["traffic"], dom.get("traffic_note")
        kernel_ms = float(band_ms.sum())
        flops_per_sample = 50.0 * sum(np.log2(n) for n in sizes) + 80.0 * n_bands
        flops_executed = sum(L["flops_executed"] for L in launches)
        calls_samples = own                                                            # samples per process_device call
        out = {
            "metric": f"stereo Msamples/sec upmixed ({n_bands}-band, STFT<={max(sizes)})",
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preheat": {"untimed_steps": preheat_steps, "ms": args.preheat_ms,
                        "first_5_steps_ms_per_step": None if cold_ms is None else round(cold_ms, 4),
                        "five_steps_after_2s_idle_ms_per_step": None if idle_ms is None else round(idle_ms, 4),
                        "software_share_of_a_cold_step_ms": None if idle_ms is None or cold_ms is None else round(cold_ms - idle_ms, 4),
                        "cold_note": "first_5 = the first calls of this process (after upx_plan_reserve: buffers and stream "
                                     "tables prepared); after_2s_idle = the same steps on an idle card with everything touched "
                                     "= the clock ramp alone; their difference is what software still adds to a cold call",
                        "why": "an idle card reaches its running clocks after ~20 steps; run before the warm-up steps, "
                               "never inside the timed region (scripts/clock_ramp_check.py, DESIGN.md 5)"},
            "kernel_timing": {"steps_with_events": timed_steps, "stride": TIMING_STRIDE,
                              "why": "launches[].ms are HIP-event averages over every 4th step of the timed region: the "
                                     "events themselves cost 1.7 % of a step (scripts/timing_cost_check.py)"},
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{cfg_name}: " + (f"{TRACKS_PER_GPU} tracks of " if batch else "") +
                            f"{seconds:g} s of {sr // 1000} kHz stereo per GPU, {n_bands} band{'s' if n_bands > 1 else ''} "
                            f"(edges {'/'.join(f'{b.f_low:g}' for b in bands)}/{bands[-1].f_high:g} Hz), STFT {sizes}, "
                            f"Blackman-Harris {100 * OVERLAP.get(args.workload, 0.75):g}% WOLA, raised-cosine crossovers XO 0.25, export Ls/C/Rs planes",
                "name": args.workload,
                "samples_per_gpu": samples_per_step_rank,
                "x_realtime": round(total_samples / sr / (elapsed / args.steps), 1),
                "parallelism": "1 GPU" if world == 1 else (
                    (f"{world} replicas, tracks rank::world, no communication" if batch else
                     f"time-sharded x{world}, one RCCL seam all-reduce per step")
                    if os.environ.get("UPX_BENCH_REHEARSAL") != "1" else f"REHEARSAL x{world} (host seam, shared device) - not a result"),
            },
            "median_kernel_ms_per_step": None if median_ms is None else round(median_ms, 4),
            "host_prep_us_per_call": None if prep_us is None else round(prep_us, 1),
            "roofline": {
                "bound": "hbm (prescribed)",
                "binding_ceiling": "fp32 vector issue (valu.executed, launches[].valu_frac) - not HBM bandwidth",
                "kernel": dom["kernel"],
                "achieved": dom["GBps"],
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": dom["frac"],
                "traffic": traffic,
                "traffic_note": traffic_note,
                "achieved_traffic_GBps": None if not traffic else round(traffic / (dom["ms"] * 1e-3) / 1e9, 1),
                "algorithmic_bytes_per_launch": dom["algo_bytes"],
                "bands_in_launch": dom["bands"],
                "avg_launch_ms": dom["ms"],
                "traffic_ratio": dom["traffic_ratio"],
                # what a pass with these kernels' lane pattern (4 bytes per lane) reaches on THIS card in THIS run (x *= 1.0f over
                # one output plane: load + store); the guide's 16-byte-per-lane copy reaches 6290 GB/s; `peak` stays its 8 TB/s
                "lane_pattern_streaming_GBps": None if stream_gbps is None else round(stream_gbps, 1),
                "guide_float4_copy_GBps": 6290.0,
                "traffic_frac_of_lane_pattern_streaming": None if not traffic or not stream_gbps else
                round(traffic / (dom["ms"] * 1e-3) / 1e9 / stream_gbps, 4),
                "limiter": "not HBM bandwidth: VALU issue + LDS exchanges at 2-4 waves per SIMD, and for the fused kernels the "
                           "in-order vector L1 (DESIGN.md 5, 8); `valu.executed` is the ceiling that binds",
            },
            "launches": launches,
            "all_bands_algorithmic_GBps": round((ALGO_BYTES_IN + ALGO_BYTES_OUT) * n_bands * calls_samples
                                                / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms > 0 else None,
            "all_bands_frac": round((ALGO_BYTES_IN + ALGO_BYTES_OUT) * n_bands * calls_samples
                                    / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if kernel_ms > 0 else None,
            "valu": {
                # what the SIMDs execute: the transforms actually run (P-point ones on the band-limited path)
                "executed": {
                    "flops_per_sample": round(flops_executed / calls_samples, 1),
                    "formula": "per launch: fused 12.5 K log2 N + 80 bands; band-limited analysis 5 K log2 P + 8 K + "
                               "80 bands P/N, synthesis 7.5 K log2 P + 9 K (K = N / hop); launches[].flops_executed",
                    "achieved": round(flops_executed / (kernel_ms * 1e-3) / 1e12, 2) if kernel_ms > 0 else None,
                    "frac": round(flops_executed / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4) if kernel_ms > 0 else None,
                },
                # the same time priced at what full-size transforms for every band would cost: a
                # speed-up figure for the pruned path, NOT a utilisation (it can exceed the peak)
                "band_equivalent": {
                    "flops_per_sample": round(flops_per_sample, 1),
                    "formula": "50 * sum_b log2 N_b + 80 * bands (SURVEY.md 8(d): algorithmic, full-size transforms)",
                    "achieved": round(flops_per_sample * calls_samples / (kernel_ms * 1e-3) / 1e12, 2) if kernel_ms > 0 else None,
                    "frac_of_peak_equivalent": round(flops_per_sample * calls_samples / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4) if kernel_ms > 0 else None,
                },
                "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            },
            "kernel_sources_sha256_16": kernel_sources_sha(),
        }
        if world == 1 and not args.no_e2e:
            try:
                if batch:
                    tracks = [synth(nominal, (4, t)) for t in range(TRACKS_PER_GPU)]
                    t0 = time.perf_counter()
                    res = plan.process_tracks(tracks)
                    first = time.perf_counter() - t0
                    del res
                    best = None
                    for _ in range(3):
                        t0 = time.perf_counter()
                        res = plan.process_tracks(tracks)
                        dt = time.perf_counter() - t0
                        del res
                        best = dt if best is None or dt < best else best
                    out["e2e"] = {"upx_process_tracks": {"ms": round(best * 1e3, 1), "tracks": TRACKS_PER_GPU,
                                                         "Msamples_per_s": round(nominal * TRACKS_PER_GPU / best / 1e6, 1),
                                                         "first_call_ms": round(first * 1e3, 1)},
                                  "note": "pageable input arrays, new result arrays per call in pooled page-locked memory "
                                          "(the first call pins the blocks); uploads, kernels and downloads of consecutive "
                                          "tracks overlap; best of 3; PCIe-inclusive (never `value`)"}
                    del tracks
                else:
                    out["e2e"] = e2e_rates(ux, plan, bands, sr, nominal, seed, fresh_process=wl_edges is None)
                    if args.workload == "c4share":
                        out["e2e"].update(e2e_multi_gpu_file(bands, sr, nominal, local_rank))
            except Exception as exc:   # a side measurement must not take the bench line down
                out["e2e"] = {"error": repr(exc)}
        else:
            out["e2e"] = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if comm is not None:
        comm.close()
    group.barrier()
    group.close()
    plan.close()


if __name__ == "__main__":
    sys.exit(main() or 0)
