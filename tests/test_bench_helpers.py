"""bench.py pieces that do not need a GPU: the PMC traffic figure is only reported for the kernel sources it was
collected from (`roofline.traffic` must be tied to the build that ran), workload table, track assignment."""
import json
import os

import bench


def test_pmc_traffic_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    sha = bench.kernel_sources_sha()
    assert len(sha) == 16 and sha == bench.kernel_sources_sha()
    root = tmp_path
    os.makedirs(root / "profiles")
    monkeypatch.setattr(bench, "ROOT", str(root))
    v, note = bench.load_pmc_traffic("k")
    assert v is None and "no profiles" in note
    rec = {"c3": {"k": {"hbm_bytes_per_launch": 123}}, "_kernel_sources_sha256_16": "0" * 16}
    (root / "profiles" / "pmc_traffic.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "kernel_sources_sha", lambda: sha)
    v, note = bench.load_pmc_traffic("k")
    assert v is None and "other kernel sources" in note            # stale summary: refuse
    rec["_kernel_sources_sha256_16"] = sha
    (root / "profiles" / "pmc_traffic.json").write_text(json.dumps(rec))
    assert bench.load_pmc_traffic("k") == (123, None)
    v, note = bench.load_pmc_traffic("other")
    assert v is None and "not in" in note
    v, note = bench.load_pmc_traffic("k", "batch")                  # counters of another workload do not transfer
    assert v is None and "no PMC passes of workload" in note


def test_committed_pmc_summary_names_every_c3_kernel():
    """The committed summary must carry the kernels the recorded workload launches (names as bench.py reports them)."""
    rec = json.load(open(os.path.join(os.path.dirname(bench.__file__), "profiles", "pmc_traffic.json")))
    assert all(w in rec for w in ("c3", "c4share", "default"))
    names = {bench.canonical_kernel_name(k) for w in ("c3", "c4share") for k in rec[w]}
    for k in ("upx_zoom_analysis_kernel<upx::ZoomCfg<8, 16, 4>>", "upx_zoom_synthesis_kernel<upx::ZoomCfg<9, 8, 4>>",
              "upx_band_kernel<upx::Cfg<10, 4, 16>, 2, false, upx::Live<0, 4>>", "upx_band_kernel<upx::Cfg<8, 4, 16>, 2, false>",
              "upx_band_kernel<upx::Cfg<11, 4, 16>, 2, false, upx::Live<0, 2>>", "upx_zoom_synthesis_kernel<upx::ZoomCfg<9, 16, 4>>"):
        assert bench.canonical_kernel_name(k) in names, k
    # rocprofv3's spelling of the general flavour and the library's own name meet
    assert bench.canonical_kernel_name("void upxk::upx_band_kernel<upx::Cfg<8, 4, 16>, 2, false, upx::Live<0, 1048576> >") == \
        bench.canonical_kernel_name("upx_band_kernel<upx::Cfg<8, 4, 16>, 2, false>")
    assert all(v["hbm_bytes_per_launch"] > 0 for w, e in rec.items() if not w.startswith("_") for v in e.values())


def test_workload_table():
    baseline = {"c1", "c2", "c3", "default", "c4share", "batch"}
    beyond = {"ov50", "ov875", "ov60", "wide65536"}          # the reference's argument space outside the BASELINE shapes
    assert set(bench.WORKLOADS) == baseline | beyond and set(bench.OVERLAP) == {"ov50", "ov875", "ov60"}
    from oracle import upmix_oracle as orc
    ov = lambda w: (lambda e, sr, m, f: orc.plan_bands(e, bench.OVERLAP.get(w, 0.75), orc.win_blackman_harris, sr,  # noqa: E731
                                                       max_block_size=m, threshold_factor=f))
    assert [(b.block_size, b.hop_size) for b in bench.workload_bands("ov875", orc.Band, ov("ov875"))][-2:] == [(1024, 128), (256, 32)]
    assert [(b.block_size, b.hop_size) for b in bench.workload_bands("ov60", orc.Band, ov("ov60"))][-1] == (256, 102)
    wide = bench.workload_bands("wide65536", orc.Band, ov("wide65536"))
    assert [(b.block_size, b.f_low, b.f_high) for b in wide] == [(65536, 0, 3000), (512, 3000, 24000.0)]
    sr, seconds, max_stft = bench.WORKLOADS["c3"][:3]
    assert (sr, seconds, max_stft) == (48000, 600, 8192)            # BASELINE configs[2]
    assert bench.WORKLOADS["c4share"][:3] == (96000, 900, 8192)     # 2 h at 96 kHz over 8 GPUs
    assert bench.TRACKS_PER_GPU * 8 == 64                           # configs[4]
    x = bench.synth(1000, 2)
    assert x.shape == (1000, 2) and x.dtype.name == "float32"


def test_named_shapes_of_configs_0_and_1():
    """--workload c1 / c2 build the plans SURVEY.md 8 states for BASELINE configs[0] / [1] (the oracle's constructors here;
    bench.py hands upmix_amd's the same arguments)."""
    from oracle import upmix_oracle as orc
    chain = lambda e, sr, m, f: orc.plan_bands(e, 0.75, orc.win_blackman_harris, sr, max_block_size=m, threshold_factor=f)  # noqa: E731
    (b,) = bench.workload_bands("c1", orc.Band, chain)
    assert (b.block_size, b.hop_size, b.f_low, b.f_high) == (2048, 512, 0.0, 24000.0)
    assert bench.WORKLOADS["c1"][:2] == (48000, 10) and bench.WORKLOADS["c1"][6] == 0          # 480 000 samples, seed 0
    c2 = bench.workload_bands("c2", orc.Band, chain)
    assert [x.block_size for x in c2] == [4096, 4096, 1024] and bench.WORKLOADS["c2"][:2] == (48000, 60)
    c3 = bench.workload_bands("c3", orc.Band, chain)
    assert [x.block_size for x in c3] == [8192, 8192, 8192, 4096, 1024, 256]


def test_cpu_baseline_states_threads_and_host(monkeypatch):
    """`cores` reported the band count as if it were the host.  `cores` is now the host's (the CPUs this
    process may run on: north_star asks for the core count of the box the CPU line was timed on), `threads_used` what ran the
    frame loops (tasks of the reference's ThreadPoolExecutor), next to cpus, model and NumPy."""
    monkeypatch.setitem(bench.WORKLOADS, "c1", (48000, 1, 2048, "configs[0], 1 s for the test", "single", 32, 0))
    line = bench.cpu_baseline("c1", target_seconds=0.5)
    assert line["kind"] == "port" and line["unit"] == "Msamples/s" and line["value"] > 0
    assert line["threads_used"] == 1                                   # one band -> one task
    assert line["cores"] == line["usable_cpus"] >= 1
    assert line["host_cpus"] == os.cpu_count() and line["numpy"] and "cpu_model" in line
    assert "WHOLE workload" in line["sample"]


def test_bare_gpus_n_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the ranks through upmix_amd.launch.run
    (fresh children of the same command) and returns their exit code - and never loads the HIP library itself."""
    import sys
    from upmix_amd import launch, _lib
    calls = []
    monkeypatch.setattr(launch, "run", lambda n, cmd, **kw: calls.append((n, cmd)) or 7)
    monkeypatch.setattr(_lib, "load", lambda: (_ for _ in ()).throw(AssertionError("the parent touched the library")))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--workload", "c2"])
    assert bench.main() == 7
    (n, cmd), = calls
    assert n == 2 and cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == sys.argv[1:]
