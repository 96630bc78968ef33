#!/usr/bin/env python3
"""profiles/r03_pmc_gemv_traffic.json from the FETCH_SIZE / WRITE_SIZE passes over tools/probe_r2.py gemv (tools/gpu_r3_pmc_gemv.sh): HBM bytes per
launch of the decode weight-stream kernels against their algorithmic weight bytes.  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section) and is doubled here."""
import json
import sys

fetch, write, out = (json.load(open(p)) for p in sys.argv[1:3]), None, sys.argv[3]
fetch = list(fetch)
fetch, write = fetch[0], fetch[1]
FP8 = len(sys.argv) > 4 and sys.argv[4] == "fp8"       # round 5: the fp8 image of the layer linears (1 byte per weight), tools/probe_r2.py gemv fp8
H, I, V = 2048, 11008, 151936
ALGO = {"gate/up": 2 * I * H * (1 if FP8 else 2), "down": H * I * (1 if FP8 else 2), "lm_head": V * H * 2}


def kind(name):                # kernel symbol -> (launch, batch) by the template arguments the decode path dispatches
    if FP8:                    # k_gemv<MODE, MT, KP = 4, STAGE, 4, F8 = true> (MT 2 = 17..32 rows, MT 1 = <= 16 rows); k_gemv32g<0, 4, 1, true> = the 4-slab down-projection at 17..32 rows
        n = name.replace(" ", "")
        if not (n.endswith(",true>") or "Lb1EEEv8GemvArgs" in n):
            return None, None
        if "k_gemv32g<0,4,1" in n or "k_gemv32gILi0ELi4ELi1E" in n or "k_gemv<0,2," in n or "k_gemvILi0ELi2E" in n:
            return "down", 32
        if "k_gemv<1,2," in n or "k_gemvILi1ELi2E" in n:
            return "gate/up", 32
        if "k_gemv<1,1," in n or "k_gemvILi1ELi1E" in n:
            return "gate/up", 1
        if "k_gemv<0,1," in n or "k_gemvILi0ELi1E" in n:
            return "down", 1
        return None, None
    if "k_gemv32ILi2E" in name or "k_gemv32<2" in name:
        return "lm_head", 32
    if "k_gemv32ILi0E" in name or "k_gemv32<0" in name:
        return "down", 32
    if "k_gemvILi1ELi2E" in name or "k_gemv<1, 2" in name:
        return "gate/up", 32
    if "k_gemvILi1ELi1E" in name or "k_gemv<1, 1" in name:
        return "gate/up", 1
    if "k_gemvILi0ELi1E" in name or "k_gemv<0, 1" in name:
        return "down", 1
    if "k_gemvILi2ELi1E" in name or "k_gemv<2, 1" in name:
        return "lm_head", 1
    return None, None


res, tot = {}, {32: [0.0, 0.0], 1: [0.0, 0.0]}
for name, c in fetch.items():
    launch, B = kind(name)
    if launch is None or "FETCH_SIZE" not in c:
        continue
    rd = c["FETCH_SIZE"]["per_dispatch"] * 1024 * 2
    wr = write.get(name, {}).get("WRITE_SIZE", {}).get("per_dispatch", 0.0) * 1024
    res[name] = {"launch": f"{launch} at batch {B}", "algorithmic_weight_bytes": ALGO[launch], "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr),
                 "traffic_over_algorithmic": round(rd / ALGO[launch], 4), "reads_plus_writes_over_weight_bytes": round((rd + wr) / ALGO[launch], 4),
                 "avg_us_under_pmc": round(c["FETCH_SIZE"]["avg_us"], 1)}
    tot[B][0] += rd
    tot[B][1] += ALGO[launch]
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python tools/probe_r2.py gemv (separate passes: the script that wrote this file -- tools/gpu_lease.sh pmc_gemv in round 5); "
                     "counter summed over its instances per dispatch, averaged over the dispatches (tools/rocpd_pmc.py, tools/gemv_traffic.py)",
           "correction": "FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE doubled (gfx950 note in MI355X_MICROARCH.md).  traffic_over_algorithmic = HBM reads / weight bytes "
                         "(as in profiles/r02_pmc_gemv_traffic.json); the writes (float32 logits of the LM head, the down-projection's four float32 slabs) are listed beside it",
           "kernels": res,
           "traffic_over_algorithmic_weighted_batch32": round(tot[32][0] / tot[32][1], 4) if tot[32][1] else None,
           "traffic_over_algorithmic_weighted_batch1": round(tot[1][0] / tot[1][1], 4) if tot[1][1] else None}, open(out, "w"), indent=1)
print(open(out).read()[:1500])
