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
    for name in ("sr_switches_reload", "sr_tail_timeouts"):
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
