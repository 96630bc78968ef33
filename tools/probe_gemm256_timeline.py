"""Where does a 256 x 256 tile's time go?  Runs the hot path's GEMM shapes on a build of gemm256.hip with -DSR_G256_TIMING
(socioreasoner_amd/libsocior_timing.so; built by tools/gpu_r4_timeline.sh, never the product library): wave 0 of every block records
the 100 MHz clock at entry, after the prologue's first wait, after the k loop, after the epilogue's stores are issued and after they are
acknowledged, plus the CU it ran on.  Prints per shape: the phases' medians, the gap between consecutive blocks on one CU, the spread of
the k-loop ends inside a round (lockstep or not) and the kernel's span."""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "socioreasoner_amd", "libsocior_timing.so"))
L.sr_op_gemm.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
L.sr_dbg_g256_times.argtypes = [C.c_void_p, C.c_int]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [("vit qkv(store)", 32768, 3840, 1280, 0, 0), ("vit proj", 32768, 1280, 1280, 1, 0), ("vit gate/up", 32768, 6912, 1280, 2, 0),
          ("vit down", 32768, 1280, 3456, 1, 0), ("lm o", 14336, 2048, 2048, 1, 0x100), ("lm gate/up", 14336, 22016, 2048, 2, 0x100),
          ("lm down", 14336, 2048, 11008, 1, 0x100), ("one round", 8192, 2048, 2048, 0, 0)]
for name, M, N, K, epi, tl in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    ldo = N // 2 if epi == 2 else N
    out = torch.zeros(M, ldo, dtype=torch.bfloat16, device="cuda")
    res = out if epi == 1 else None
    nb = ((M + 255) // 256) * (N // 256)
    for _ in range(3):
        assert L.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), ldo, None, P(res), None, epi | 0x200 | tl, s) == 0
    torch.cuda.synchronize()
    t = np.zeros((nb, 6), np.int64)
    assert L.sr_dbg_g256_times(t.ctypes.data, nb) == 0
    us = lambda x: x * 0.01
    t0 = t[:, 0].min()
    pro, loop, epi_i, epi_d, tot = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2]), us(t[:, 5] - t[:, 3]), us(t[:, 5] - t[:, 0])
    cu = ((t[:, 4] >> 32) & 0xF) << 16 | (t[:, 4] & 0xFF00)          # XCC_ID | HW_ID's se_id, sh_id, cu_id
    order = np.argsort(t[:, 0])
    gaps, per_cu = [], {}
    for b in order:
        per_cu.setdefault(int(cu[b]), []).append(b)
    for k, bl in per_cu.items():
        for x, y in zip(bl[:-1], bl[1:]):
            gaps.append(us(t[y, 0] - t[x, 5]))
    rounds = {}
    for k, bl in per_cu.items():
        for r, b in enumerate(bl):
            rounds.setdefault(r, []).append(us(t[b, 2] - t0))
    spread = {r: round(float(np.std(v)), 2) for r, v in rounds.items() if len(v) > 32}
    med = lambda v: round(float(np.median(v)), 2)
    row = {"shape": name, "M": M, "N": N, "K": K, "k_tiles": K // 64, "tiles": nb, "cus_seen": len(per_cu), "span_us": round(us(t[:, 5].max() - t0), 1),
           "prologue_us": med(pro), "k_loop_us": med(loop), "k_loop_us_per_k_tile": round(med(loop) / (K // 64), 3), "epilogue_issue_us": med(epi_i),
           "epilogue_drain_us": med(epi_d), "block_us": med(tot), "gap_between_blocks_on_a_cu_us": med(gaps) if gaps else None,
           "gap_p90": round(float(np.percentile(gaps, 90)), 2) if gaps else None,
           "prologue_first_round_us": med(pro[order[:256]]), "prologue_later_us": med(pro[order[256:]]) if nb > 256 else None,
           "k_loop_first_round_us": med(loop[order[:256]]), "k_loop_later_us": med(loop[order[256:]]) if nb > 256 else None,
           "std_of_k_loop_end_by_round_us": spread}
    xcc = (t[:, 4] >> 32) & 0xF
    row["per_xcd"] = {int(x): {"tiles": int((xcc == x).sum()), "block_us_mean": round(float(tot[xcc == x].mean()), 2), "k_loop_us_mean": round(float(loop[xcc == x].mean()), 2),
                               "last_end_us": round(float(us(t[xcc == x, 5].max() - t0)), 1)} for x in sorted(set(xcc.tolist()))}
    # gap by round (per CU: the idle time in front of its r-th block)
    gr = {}
    for k, bl in per_cu.items():
        for r, (x, y) in enumerate(zip(bl[:-1], bl[1:])):
            gr.setdefault(r + 1, []).append(us(t[y, 0] - t[x, 5]))
    row["gap_by_round_us"] = {r: round(float(np.median(v)), 2) for r, v in gr.items() if len(v) > 32}
    print(json.dumps(row), flush=True)
