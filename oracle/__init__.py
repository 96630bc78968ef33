"""CPU oracle for the SocioReasoner inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: it may be
imported by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg, and by nothing in ``socioreasoner_amd/`` or ``roll/``.

Pinning status: the reference (AMAP-ML/SocioReasoner) ships no tests or golden
vectors for this path.  The arithmetic lives in the
un-vendored third-party dependency ``transformers`` (Qwen2.5-VL), so this oracle
restates HF's bf16-eager semantics and is pinned against outputs of that
dependency generated in the build container (``tools/make_golden.py`` ->
``tests/golden/``), and against the reference's own pure functions executed in
the build container for the parser / raster / layout rows.

Parity UNPINNED (no reference output exists to pin against; the definition is
this repo's own and is stated where it lives): the fp8 weight mode of
BASELINE.json configs[4] (``model_ref.QuantW``), ``cv2.INTER_NEAREST`` and the
SAM2 forward (restated from documentation / stood in for), and the random
stream of the sampler (vLLM's cannot be reproduced; the sampling tests are
distributional).
"""
