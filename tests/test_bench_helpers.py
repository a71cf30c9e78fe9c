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
    assert set(bench.WORKLOADS) == {"c3", "default", "c4share", "batch"}
    sr, seconds, max_stft, _ = bench.WORKLOADS["c3"]
    assert (sr, seconds, max_stft) == (48000, 600, 8192)            # BASELINE configs[2]
    assert bench.WORKLOADS["c4share"][:3] == (96000, 900, 8192)     # 2 h at 96 kHz over 8 GPUs
    assert bench.TRACKS_PER_GPU * 8 == 64                           # configs[4]
    x = bench.synth(1000, 2)
    assert x.shape == (1000, 2) and x.dtype.name == "float32"
