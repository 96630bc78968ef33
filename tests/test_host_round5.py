"""Round 5 host-side tests (CPU): what a checkpoint's config.json may ask of the engine, where a model_name_or_path points on this machine,
the dispatcher's reaction to another rank's abort, the scheduler's host-time bookkeeping."""
import json
import os
import threading
import time
import warnings

import pytest


# ------------------------------------------------------------------------------------------------ unsupported checkpoint geometries are refused by name
def test_checkpoint_geometry_is_checked_against_the_engines_limits():
    """geometry_from_hf_config used to accept any Qwen2.5-VL config.json: an UNTIED LM head (Qwen2.5-VL-7B) would silently have been replaced
    by the embedding matrix (sr_load_weight ignores lm_head.weight), head_dim != 128 and hidden > 2048 failed at the first decode.  Now refused
    up front, every reason named; the 3B and the tiny geometry pass."""
    from socioreasoner_amd.config import check_supported, geometry_3b, geometry_from_hf_config, geometry_tiny, geometry_to_hf_config
    check_supported(geometry_3b())
    check_supported(geometry_tiny())
    cfg = geometry_to_hf_config(geometry_3b())
    assert geometry_from_hf_config(cfg) == geometry_3b()
    seven_b = dict(cfg, hidden_size=3584, num_attention_heads=28, num_key_value_heads=4, intermediate_size=18944, tie_word_embeddings=False,
                   vision_config=dict(cfg["vision_config"], out_hidden_size=3584))
    with pytest.raises(ValueError) as ei:
        geometry_from_hf_config(seven_b)
    msg = str(ei.value)
    assert "tie_word_embeddings" in msg and "hidden_size 3584" in msg
    with pytest.raises(ValueError, match="head_dim 64"):
        geometry_from_hf_config(dict(cfg, num_attention_heads=32, num_key_value_heads=4))            # 2048 / 32 = 64
    with pytest.raises(ValueError, match="GQA group"):
        geometry_from_hf_config(dict(cfg, head_dim=128, num_attention_heads=16, num_key_value_heads=3))
    with pytest.raises(ValueError, match="mrope_section"):
        geometry_from_hf_config(dict(cfg, rope_scaling={"type": "mrope", "mrope_section": [16, 24, 16]}))


# ------------------------------------------------------------------------------------------------ one checkpoint policy, hub ids through the local HF cache
def test_model_path_resolution_policy(tmp_path, monkeypatch):
    """socioreasoner_amd.checkpoints.resolve: synthetic:* | an existing directory | a hub id found in the LOCAL HuggingFace cache (never fetched) |
    else FileNotFoundError -- or, with SR_ALLOW_SYNTHETIC_WEIGHTS=1, a loud fallback to synthetic weights.  The LM strategy and seg_infer's provider
    both go through it (the shipped YAML names hub ids, as the reference's does: examples/infer/rlvr_megatron.yaml:9, 41)."""
    from socioreasoner_amd import checkpoints
    monkeypatch.delenv("SR_ALLOW_SYNTHETIC_WEIGHTS", raising=False)
    assert checkpoints.resolve("", "x", "synthetic:3b") == ("synthetic", "synthetic:3b")
    assert checkpoints.resolve("synthetic:tiny", "x", "synthetic:3b") == ("synthetic", "synthetic:tiny")
    d = tmp_path / "ckpt"
    d.mkdir()
    assert checkpoints.resolve(str(d), "x", "synthetic:3b") == ("dir", str(d))
    # a hub id with a snapshot in the local cache (the layout huggingface_hub writes: models--org--name/{refs/main, snapshots/<commit>/...})
    hub = tmp_path / "hf" / "hub"
    snap = hub / "models--acme--tiny-model" / "snapshots" / "0123456789abcdef0123456789abcdef01234567"
    snap.mkdir(parents=True)
    (snap / "config.json").write_text("{}")
    refs = hub / "models--acme--tiny-model" / "refs"
    refs.mkdir()
    (refs / "main").write_text("0123456789abcdef0123456789abcdef01234567")
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    monkeypatch.setenv("HF_HUB_CACHE", str(hub))
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    import huggingface_hub.constants as hc
    monkeypatch.setattr(hc, "HF_HUB_CACHE", str(hub), raising=False)
    kind, path = checkpoints.resolve("acme/tiny-model", "x", "synthetic:3b")
    assert kind == "dir" and os.path.samefile(path, snap)
    with pytest.raises(FileNotFoundError, match="local HuggingFace cache"):
        checkpoints.resolve("acme/not-here", "actor_infer", "synthetic:3b")
    with pytest.raises(FileNotFoundError):
        checkpoints.resolve("/no/such/dir", "seg_infer", "synthetic:sam2-hiera-large")
    monkeypatch.setenv("SR_ALLOW_SYNTHETIC_WEIGHTS", "1")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert checkpoints.resolve("acme/not-here", "actor_infer", "synthetic:3b") == ("synthetic-fallback", "synthetic:3b")
    assert any("SYNTHETIC" in str(x.message) for x in w)


# ------------------------------------------------------------------------------------------------ the dispatcher thread stops when another rank aborts
def test_dispatcher_stops_dealing_when_another_rank_aborted():
    """rank 0's dispatcher thread waits for a worker to drop below its request cap; when another rank's failure path sets `abort` it used to
    spin until the round's timeout (3600 s).  Now the cap-wait loop polls aborted(): it ends within a few polls and records the error."""
    from torch.distributed import HashStore
    from socioreasoner_amd.dispatch import CrossRankDispatcher
    d = CrossRankDispatcher(HashStore(), rank=0, world=2, round_id=1, max_running_requests=1, poll_s=0.002, timeout_s=60.0)
    t = threading.Thread(target=d._dispatch, args=([2, 2],), daemon=True)       # 4 requests, cap 1 per worker: the third must wait for a `done`
    t.start()
    time.sleep(0.1)
    assert t.is_alive()                      # two dealt, waiting below the cap
    d._abort()                               # what any rank's failure path does
    t.join(timeout=5.0)
    assert not t.is_alive(), "the dispatcher kept waiting after the abort"
    assert d.errors and "aborted" in str(d.errors[0])


# ------------------------------------------------------------------------------------------------ the library's switch table and the new entry points are declared
def test_round5_entry_points_are_declared_and_bound():
    from socioreasoner_amd import lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "socior.h")).read()
    for name in ("sr_switches_reload",):
        assert name in hdr and name in lib.SIGNATURES
    L = lib.load()
    assert L.sr_switches_reload() == 0      # (no GPU needed: it only reads the environment)


def test_gpu_lease_script_parses_and_documents_its_stages():
    """tools/gpu_lease.sh is the one script behind every number under profiles/r05_*: it must parse, and every stage it can run is named in its header
    (and the other way round) so that the list a reader sees is the list that exists."""
    import re
    import subprocess
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gpu_lease.sh")
    assert subprocess.run(["bash", "-n", path]).returncode == 0
    text = open(path).read()
    head = text.split('cd "$(dirname "$0")/.."')[0]
    documented = set()
    for line in head.splitlines():
        m = re.match(r"#\s{3}([a-z0-9_/ |]+?)\s{2,}\S", line)
        if m:
            documented.update(s for s in re.split(r"[\s/|]+", m.group(1)) if s)
    stages = set()
    for m in re.finditer(r"^\s{4}([a-z0-9_|]+)\)", text, re.M):
        stages.update(m.group(1).split("|"))
    assert stages, "no stages found"
    assert stages <= documented, sorted(stages - documented)


def _device_disassembly():
    """disassembly of every gfx950 code object bundled into libsocior.so, keyed by kernel symbol (llvm-objcopy / clang-offload-bundler / llvm-objdump of the
    ROCm toolchain; the .hip_fatbin section is a sequence of offload bundles, one per translation unit)"""
    import re
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "socioreasoner_amd", "libsocior.so")
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not os.path.exists(so) or not all(os.path.exists(t) for t in tools):
        pytest.skip("needs the built libsocior.so and the ROCm LLVM tools")
    funcs = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", so, os.path.join(td, "copy.so")], check=True)
        data = open(fat, "rb").read()
        idx = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
        for i, a in enumerate(idx):
            b = idx[i + 1] if i + 1 < len(idx) else len(data)
            piece, co = os.path.join(td, f"b{i}.bin"), os.path.join(td, f"d{i}.co")
            open(piece, "wb").write(data[a:b])
            subprocess.run([tools[1], "--unbundle", "--type=o", f"--input={piece}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
            name = None
            for line in subprocess.run([tools[2], "-d", co], check=True, capture_output=True, text=True).stdout.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
                if m:
                    name = m.group(1)
                    funcs[name] = []
                elif name and line.strip():
                    funcs[name].append(line.split("//")[0].strip())
    return funcs


def test_built_library_keeps_the_two_compiler_findings_of_round_5():
    """Round 5 found two places where hipcc's code was not what the source promised; both fixes live in how the source is WRITTEN, so a toolchain or source
    change could silently undo them.  This reads the ISA of the built library:
      * k_attn_prefill2 (hand-issued V^T reads): plain ds_read_b64, never the paired ds_read2st64_b64 (half rate, 32-bank rule: 2-way conflicts), and no
        instruction touches a read's destination registers between the read and the counted lgkmcnt wait that covers it (a copy there would copy stale data);
      * the decode GEMVs' ring loops (k_gemv32 / k_gemv at <= 32 rows): counted `s_waitcnt vmcnt(N)` between the MFMAs -- with a refill behind a condition
        hipcc waits vmcnt(0) at the top of every round."""
    import re
    funcs = _device_disassembly()

    def one(pattern):
        names = [n for n in funcs if re.search(pattern, n)]
        assert len(names) == 1, (pattern, names)
        return funcs[names[0]]

    for pat in (r"k_attn_prefill2ILi128ELb1ELi2ELi2ELb1E", r"k_attn_prefill2ILi80ELb0ELi8ELi3ELb1E"):
        body = one(pat)
        text = "\n".join(body)
        assert "ds_read2st64_b64" not in text and "ds_read2_b64" not in text, pat
        reads = [i for i, l in enumerate(body) if l.startswith("ds_read_b64 ")]
        assert len(reads) >= 20, (pat, len(reads))
        for i in reads:
            m = re.match(r"ds_read_b64 v\[(\d+):(\d+)\]", body[i])
            regs = {int(m.group(1)), int(m.group(2))}
            for l in body[i + 1:]:
                if l.startswith("s_waitcnt") and "lgkmcnt" in l:
                    break
                used = set()
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", l):
                    used.update(range(int(a), int(b) + 1))
                used.update(int(x) for x in re.findall(r"\bv(\d+)\b", l))
                if l.startswith("ds_read_b64 "):      # another hand-issued read: only its ADDRESS register may coincide (never with a pending destination)
                    addr = re.match(r"ds_read_b64 v\[\d+:\d+\], v(\d+)", l)
                    used = {int(addr.group(1))}
                assert not (used & regs), (pat, body[i], l)
    # round 6: the LDS-staged launches of <= 4 rows too (gate/up with its RMSNorm prologue, the LM head) and the row-major down-projection at batch 1
    for pat, floor in ((r"k_gemv32ILi0ELi4ELb1E", 16), (r"k_gemv32ILi2ELi1ELb1E", 16), (r"k_gemvILi1ELi2ELi4ELb0ELi4ELb0E", 16), (r"k_gemvILi3ELi2ELi4ELb0ELi4ELb0E", 16),
                       (r"k_gemvILi1ELi1ELi4ELb1ELi4ELb0E", 12), (r"k_gemvILi2ELi1ELi1ELb1ELi4ELb0E", 12), (r"k_gemvILi0ELi1ELi4ELb0ELi4ELb0E", 16)):
        body = one(pat)
        counts = {int(x) for l in body for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", l)}
        assert len([c for c in counts if c >= 8]) >= 3 and max(counts) >= floor, (pat, sorted(counts))
