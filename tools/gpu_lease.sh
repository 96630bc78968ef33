#!/bin/bash
# ONE parametrised script for everything this repo runs on a leased MI355X (replaces the 35 one-off tools/gpu_r3_*.sh / gpu_r4_*.sh / _run_*.sh
# lease scripts of rounds 1-4, which stay in the git history):
#     gpurun --timeout 1500 -- 'bash tools/gpu_lease.sh <stage> [<stage> ...]'
# Every stage writes under gpurun_out/ (merged back by gpurun); summaries worth judging are copied into profiles/ by hand afterwards.
# Stages:
#   tests5      tests/test_gpu_round5.py
#   suite       the whole -m gpu suite
#   trace_s32   rocprofv3 kernel trace of the static 32-row bench            -> gpurun_out/${RT}_bench_s32_kernel_stats.md
#   trace_c32   ... of the headline (continuous) configuration               -> gpurun_out/${RT}_bench_c32_kernel_stats.md
#   trace_b1    ... of batch 1                                               -> gpurun_out/${RT}_bench_b1_kernel_stats.md
#   trace_fp8   ... of --fp8 static 32 rows                                  -> gpurun_out/${RT}_bench_fp8_s32_kernel_stats.md
#   pmc_gemm / pmc_attn / pmc_sam2   SQ (+ FETCH_SIZE) passes of the GEMM / prefill-attention / SAM2 float32 kernels -> gpurun_out/${RT}_pmc_gemm256.json / ${RT}_pmc_attn_prefill.json / ${RT}_pmc_sam2_f32.json
#   pmc_gemv    FETCH_SIZE / WRITE_SIZE passes of the decode weight stream (bf16 and fp8) -> gpurun_out/${RT}_pmc_gemv_traffic*.json
#   bench       the driver's command (python bench.py)                       -> gpurun_out/${RT}_bench_default_line.json
#   configs     the other configurations of README (pair, fp8, fp8-mx 896, 64 / 128 rows, no-overlap, batch 1), 2 steps each
#   smoke       __graft_entry__.smoke()
#   fp8tests    the fp8 GPU tests only
#   sam2tests   the SAM2 / float32 GEMM GPU tests only
#   trace_sam2  kernel trace of the SAM2 float32 encoder (tools/prof_sam2_encoder.py) -> gpurun_out/${RT}_sam2_f32_*.md
#   bench_pmc   one bench step with the in-run rocprofv3 PMC passes (roofline.traffic measured, not a file ratio)
#   pipeline    tools/run_example_small.py with 4 scripted objects per stage, SAM2 float32 / bf16 / no answers -> gpurun_out/${RT}_pipeline_*.json
#   pipeline250  the two-stage pipeline at the reference's scale, streamed (default) against SOCIOSEG_STREAM=0, twice each -> gpurun_out/${RT}_pipeline250_ab.txt
#   pipetests   the GPU tests that run the pipeline / the request loop
#   gemm_f32    tools/bench_gemm_f32.py: the split-bf16 float32 GEMM against the f32-input MFMA kernel  -> gpurun_out/${RT}_gemm_f32_split.jsonl
#   sam2bench   tools/bench_sam2_modes.py (float32 split / f32-input / bf16)  -> gpurun_out/${RT}_sam2_modes.json
#   host_scaling    1 / 2 / 4 / 8 gloo ranks on the one device: host ms per scheduling round -> gpurun_out/${RT}_host_scaling.jsonl
#   sched_ab    admission share from measured costs (default) against the 32-row table (SR_SCHED_ONLINE=0) at 32 / 64 / 128 rows, twice each
#   counted_ab  batch-1 and static 32-row bench with SR_GEMV_COUNTED=0 / 1, alternating twice (round 6: counted loops for the staged <= 4-row GEMVs)
#   tests6      tests/test_gpu_round6.py (TESTS6_K selects)
#   xlds_ab     the headline and the static 32-row bench with SR_GEMV_XLDS=0 / 1 / 3 (XLDS_SET selects; x-stationary gate/up and down-projection GEMVs, + 8 / 16: ln2 inside the gate/up launch), alternating twice -> gpurun_out/${RT}_gemv_xlds_ab.txt
#   masked_trace  kernel traces of 63 decode steps on the ordinary stream and on CU-masked streams (8 / 7 / 5 of 8 CUs per shader engine) -> gpurun_out/${RT}_masked_trace.txt
#   masked_trace_px  the 7 / 5-of-8 traces again with the scheduler's CU hint (sr_rows_set_cus: x-stationary gate/up and down GEMVs) -> gpurun_out/${RT}_masked_trace_px.txt
#   attn_ab     prefill attention with hand-issued V^T reads (default) against SR_ATTN_VASM=0, kernel trace, twice each -> gpurun_out/${RT}_attn_vasm_ab.txt
#   pmc_lds_all     LDS bank conflicts of every kernel of the bench and of the SAM2 float32 encoder -> gpurun_out/${RT}_pmc_lds_all.txt
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
RT=${RT:-r06}      # round tag of every output file (profiles/r06_*)
QUIET="--no-cpu-baseline --no-latency --no-sam --no-more-rows"

line() { python - "$1" "$2" <<'EOF'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
except Exception as e:
    print(sys.argv[2], "NO JSON LINE", e); sys.exit(0)
r, st = d.get("roofline") or {}, d.get("static_batch") or {}
print(sys.argv[2], "value", d["value"], d["unit"], "| ms/step", d["ms_per_step"], "| decode step ms", r.get("decode_step_ms"), "| gemv frac", r.get("frac"),
      "| phases", {k: v for k, v in d["phase_ms_per_step"].items() if not isinstance(v, dict)}, "| checksum", d["result_checksum"])
EOF
}

trace() {   # trace <name> <bench args...>: kernel trace + stats summary
  local n=$1; shift
  rm -rf /tmp/prof_${RT}_$n
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_${RT}_$n -o $n -- python $R/bench.py "$@" > $R/gpurun_out/${RT}_prof_$n.log 2>&1; echo "trace $n exit $?")
  local DB=$(find /tmp/prof_${RT}_$n -name "${n}_results.db" | head -1)
  rm -f gpurun_out/${RT}_bench_${n}_kernel_stats.md
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/${RT}_bench_${n}_kernel_stats.md > /dev/null && head -16 gpurun_out/${RT}_bench_${n}_kernel_stats.md | cut -c1-170
  [ -n "$DB" ] && python tools/rocpd_gaps.py $DB > gpurun_out/${RT}_gaps_$n.json 2>/dev/null
  line gpurun_out/${RT}_prof_$n.log "traced $n:"
  rm -rf /tmp/prof_${RT}_$n
}

for stage in "$@"; do
  echo "================ stage $stage ($(date +%T))"
  case $stage in
    tests6) timeout 2400 python -m pytest tests/test_gpu_round6.py -q -m gpu ${TESTS6_K:+-k "$TESTS6_K"} 2>&1 | tail -15 ;;
    xlds_ab) for rep in 1 2; do for v in ${XLDS_SET:-0 1 3}; do
        SR_GEMV_XLDS=$v timeout 600 python bench.py --steps 3 --warmup 1 $QUIET --no-pmc > gpurun_out/${RT}_xl_c32_$v.log 2>&1; line gpurun_out/${RT}_xl_c32_$v.log "headline SR_GEMV_XLDS=$v rep $rep:"
        SR_GEMV_XLDS=$v timeout 600 python bench.py --static --steps 3 --warmup 1 $QUIET --no-pmc > gpurun_out/${RT}_xl_s32_$v.log 2>&1; line gpurun_out/${RT}_xl_s32_$v.log "static 32 SR_GEMV_XLDS=$v rep $rep:"
      done; done | tee gpurun_out/${RT}_gemv_xlds_ab.txt ;;
    counted_ab) for rep in 1 2; do for v in 0 1; do
        SR_GEMV_COUNTED=$v timeout 600 python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-sam --no-pmc > gpurun_out/${RT}_cnt_b1_$v.log 2>&1; line gpurun_out/${RT}_cnt_b1_$v.log "batch 1 SR_GEMV_COUNTED=$v rep $rep:"
        SR_GEMV_COUNTED=$v timeout 600 python bench.py --static --steps 3 --warmup 1 $QUIET --no-pmc > gpurun_out/${RT}_cnt_s32_$v.log 2>&1; line gpurun_out/${RT}_cnt_s32_$v.log "static 32 SR_GEMV_COUNTED=$v rep $rep:"
      done; done | tee gpurun_out/${RT}_gemv_counted_ab.txt ;;
    masked_trace|masked_trace_px)      # (masked_trace: the streaming GEMVs on every stream; masked_trace_px: the scheduler's CU hint on the masked ones -> the x-stationary GEMVs)
      [ $stage = masked_trace ] && export PROBE_NO_HINT=1 || export PROBE_NO_HINT=0
      for w in $([ $stage = masked_trace ] && echo plain m8 m7 m5 || echo m7 m5); do rm -rf /tmp/mt_$w
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/mt_$w -o t -- python $R/tools/probe_masked_trace.py $w 2>/dev/null | grep "decode ms")
        DB=$(find /tmp/mt_$w -name "t_results.db" | head -1)
        [ -n "$DB" ] && python tools/rocpd_stats.py $DB /tmp/mt_$w/stats.md > /dev/null && grep "k_gemv\|k_rmsnorm_row\|k_attn_dec\|k_step" /tmp/mt_$w/stats.md | awk -F"|" -v w=$w '{printf "%s %s calls %s avg_us %s\n", w, substr($2,1,60), $3, $5}'
        [ -n "$DB" ] && python tools/rocpd_gaps.py $DB 2>/dev/null | head -c 600; echo
      done | tee gpurun_out/${RT}_${stage}.txt ;;
    tests5) timeout 2400 python -m pytest tests/test_gpu_round5.py -q -m gpu ${TESTS5_K:+-k "$TESTS5_K"} 2>&1 | tail -15 ;;
    suite)  timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ;;
    fp8tests) timeout 1500 python -m pytest tests -q -m gpu -k "f8 or fp8" 2>&1 | tail -8 ;;
    pipeline) for dt in float32 bf16; do SR_SAM2_DTYPE=$dt SCRIPTED_OBJECTS=4 OUT=/tmp/example_out SOCIOSEG_NUM_SAMPLES=64 timeout 900 python tools/run_example_small.py 2> gpurun_out/${RT}_pipeline_$dt.err | tail -1 | tee gpurun_out/${RT}_pipeline_$dt.json | cut -c1-900; done
              SCRIPTED_OBJECTS=0 OUT=/tmp/example_out SOCIOSEG_NUM_SAMPLES=64 timeout 900 python tools/run_example_small.py 2>/dev/null | tail -1 | tee gpurun_out/${RT}_pipeline_noanswers.json | cut -c1-600 ;;
    pipeline250)   # the two-stage pipeline at the reference's scale (250 samples, rollout_batch_size 250, YAML sampling, 4 scripted objects per stage): streamed against the batch order, twice each
      for rep in 1 2; do for st in 1 0; do
        SOCIOSEG_STREAM=$st SCRIPTED_OBJECTS=4 OUT=/tmp/example_out SOCIOSEG_NUM_SAMPLES=250 ROLLOUT_BATCH=250 timeout 900 python tools/run_example_small.py 2> gpurun_out/${RT}_pipeline250_s$st.err | tail -1 > gpurun_out/${RT}_pipeline250_s$st.json
        python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/${RT}_pipeline250_s$st.json')); print('SOCIOSEG_STREAM=$st rep $rep', {k: d[k] for k in ('streamed','samples_per_s','run_s','giou_acc','files','wall_s_by_phase')}); print('   ', [{k: g.get(k) for k in ('requests','served_as','steps','steps_shared','admissions','rounds','host_ms','poll_wait_ms')} for g in d['generate_calls']])
except Exception as e:
    print('SOCIOSEG_STREAM=$st rep $rep FAILED', e); print(open('gpurun_out/${RT}_pipeline250_s$st.err').read()[-3000:])
"
      done; done | tee gpurun_out/${RT}_pipeline250_ab.txt ;;
    pipetests) timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_sam2.py tests/test_gpu_round4.py -q -m gpu -x -k "pipeline or serving or request or level" 2>&1 | tail -15 ;;
    trace_sam2) rm -rf /tmp/prof_${RT}_sam; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${RT}_sam -o sam -- python $R/tools/prof_sam2_encoder.py f32 > $R/gpurun_out/${RT}_prof_sam2.log 2>&1; echo "trace sam2 exit $?")
                DB=$(find /tmp/prof_${RT}_sam -name "sam_results.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/${RT}_sam2_f32_encoder_kernel_stats.md > /dev/null && head -20 gpurun_out/${RT}_sam2_f32_encoder_kernel_stats.md | cut -c1-170
                [ -n "$DB" ] && python tools/rocpd_by_grid.py $DB gpurun_out/${RT}_sam2_f32_by_grid.md 30 > /dev/null 2>&1 && head -36 gpurun_out/${RT}_sam2_f32_by_grid.md | cut -c1-200 ;;
    bench_pmc) timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sam --no-more-rows --no-pipeline > gpurun_out/${RT}_bench_pmc.log 2> gpurun_out/${RT}_bench_pmc.err; python -c "
import json; d=json.loads([l for l in open('gpurun_out/${RT}_bench_pmc.log') if l.startswith('{')][-1]); r=d['roofline']; print({k: r[k] for k in ('frac','traffic','traffic_over_algorithmic','bytes_per_launch')}); print(r['traffic_source'][:160]); print(d['latency_b1']['roofline'])" ;;
    host_scaling) : > gpurun_out/${RT}_host_scaling.jsonl
      for n in 1 2 4 8; do
        SR_DIST_BACKEND=gloo timeout 900 python bench.py --gpus $n --batch 8 --waves 2 --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-sam --no-more-rows > gpurun_out/${RT}_hs.log 2> gpurun_out/${RT}_hs.err
        python - $n <<'PY' | tee -a gpurun_out/${RT}_host_scaling.jsonl
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/${RT}_hs.log") if l.startswith("{")][-1])
    sc = d["phase_ms_per_step"]["scheduler"]
    print(json.dumps({"ranks_on_one_device": int(sys.argv[1]), "n_gpus": d["n_gpus"], "nranks": d["config"]["exchange"]["nranks"], "tiles_per_s_all_ranks": d["value"], "ms_per_step": d["ms_per_step"],
                      "host_threads_per_rank": d["host_threads_per_rank"], "rounds_per_step": sc["rounds"], "host_ms_per_round": sc["host_ms_per_round"], "poll_wait_ms_per_round": sc["poll_wait_ms_per_round"]}))
except Exception as e:
    print(json.dumps({"ranks_on_one_device": int(sys.argv[1]), "error": str(e)[:200]}))
PY
      done ;;
    sched_ab) for cfg in "--batch 32" "--batch 64" "--batch 128"; do for rep in 1 2; do for v in 0 1; do
        SR_SCHED_ONLINE=$v timeout 900 python bench.py $cfg --steps 3 --warmup 1 $QUIET > gpurun_out/${RT}_cfg.log 2> gpurun_out/${RT}_cfg.err
        python - "$cfg online=$v rep $rep" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/${RT}_cfg.log") if l.startswith("{")][-1]); sc = d["phase_ms_per_step"]["scheduler"]
print(sys.argv[1], "|", d["value"], "tiles/s | ms/step", d["ms_per_step"], "| shares", sc["admit_cus_per_se"], "| decode step shared", sc["decode_step_ms_shared"])
PY
      done; done; done ;;
    gemm_f32) timeout 600 python tools/bench_gemm_f32.py | tee gpurun_out/${RT}_gemm_f32_split.jsonl ;;
    sam2bench) timeout 900 python tools/bench_sam2_modes.py | tee gpurun_out/${RT}_sam2_modes.json ;;
    sam2tests) timeout 1500 python -m pytest tests/test_gpu_sam2.py tests/test_gpu_round4.py -x -q -m gpu -k "sam2 or gemm_f32 or seg_infer" 2>&1 | tail -8 ;;
    smoke)  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
    trace_s32) trace s32 --static --steps 2 --warmup 1 $QUIET ;;
    trace_c32) trace c32 --steps 2 --warmup 1 $QUIET ;;
    trace_b1)  trace b1 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-sam ;;
    trace_fp8) trace fp8_s32 --fp8 --static --steps 2 --warmup 1 $QUIET ;;
    attn_ab)   # prefill attention with the V^T reads issued by hand (default) / left to the compiler, kernel trace of tools/probe_attn.py, twice each
      for rep in 1 2; do for v in 0 1; do rm -rf /tmp/attn_ab
        (cd /tmp && SR_ATTN_VASM=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/attn_ab -o ab -- python $R/tools/probe_attn.py > $R/gpurun_out/${RT}_attn_ab.log 2>&1)
        DB=$(find /tmp/attn_ab -name "ab_results.db" | head -1)
        [ -n "$DB" ] && python tools/rocpd_stats.py $DB /tmp/attn_ab/stats.md > /dev/null && grep "attn" /tmp/attn_ab/stats.md | sed "s/^/SR_ATTN_VASM=$v rep $rep /" | cut -c1-200
      done; done | tee gpurun_out/${RT}_attn_vasm_ab.txt ;;
    pmc_lds_all)   # LDS bank conflicts of EVERY kernel of the headline bench and of the SAM2 float32 encoder: which ones have any
      for w in bench sam2; do rm -rf /tmp/pmc_lds
        if [ $w = bench ]; then CMD="python $R/bench.py --steps 1 --warmup 1 $QUIET --no-pmc --no-pipeline"; else CMD="python $R/tools/prof_sam2_encoder.py f32"; fi
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d /tmp/pmc_lds -o p -- $CMD > $R/gpurun_out/${RT}_pmc_lds_$w.log 2>&1)
        db=$(find /tmp/pmc_lds -name "*.db" | head -1)
        [ -n "$db" ] && python tools/rocpd_pmc.py "$db" gpurun_out/${RT}_pmc_lds_$w.json > /dev/null
        python - gpurun_out/${RT}_pmc_lds_$w.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
rows = []
for k, v in d.items():
    a = v.get("SQ_LDS_IDX_ACTIVE", {}).get("per_dispatch", 0)
    if a > 0:
        c = v.get("SQ_LDS_BANK_CONFLICT", {}).get("per_dispatch", 0)
        rows.append((c / a, k[:90], round(a), round(c), round(v["SQ_LDS_IDX_ACTIVE"]["avg_us"], 1), v["SQ_LDS_IDX_ACTIVE"].get("dispatches")))
for r in sorted(rows, reverse=True):
    print("%.3f" % r[0], *r[1:])
PY
      done | tee gpurun_out/${RT}_pmc_lds_all.txt ;;
    pmc_gemm|pmc_attn|pmc_sam2)   # SQ / fetch counters of the GEMM (tools/probe_r2.py gemm) or prefill-attention (tools/probe_attn.py) kernels: one pass per group
      if [ $stage = pmc_gemm ]; then PROBE="tools/probe_r2.py gemm"; OUT=${RT}_pmc_gemm256${PMC_TAG}.json; MATCH="gemm"
        CGRP="FETCH_SIZE|SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES|SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY|SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16"
      elif [ $stage = pmc_sam2 ]; then PROBE="tools/prof_sam2_encoder.py f32"; OUT=${RT}_pmc_sam2_f32${PMC_TAG}.json; MATCH="f32"
        CGRP="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES|SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY|SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS|SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
      else PROBE="tools/probe_attn.py"; OUT=${RT}_pmc_attn_prefill${PMC_TAG}.json; MATCH="attn"
        CGRP="FETCH_SIZE|SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY|SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES|SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU|SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE|SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
      fi
      rm -rf gpurun_out/pmc_${RT}_$stage; mkdir -p gpurun_out/pmc_${RT}_$stage
      i=0; IFS='|'; for g in $CGRP; do unset IFS; i=$((i+1)); rm -rf /tmp/pmc_g$i
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $g -d /tmp/pmc_g$i -o p -- python $R/$PROBE > $R/gpurun_out/pmc_${RT}_$stage/pass$i.log 2>&1)
        db=$(find /tmp/pmc_g$i -name "*.db" | head -1)
        [ -n "$db" ] && python tools/rocpd_pmc.py "$db" gpurun_out/pmc_${RT}_$stage/pass$i.json > /dev/null 2>> gpurun_out/pmc_${RT}_$stage/pass$i.log
        rm -rf /tmp/pmc_g$i; IFS='|'; done; unset IFS
      python - gpurun_out/pmc_${RT}_$stage gpurun_out/$OUT $MATCH "$PROBE" <<'PY'
import glob, json, sys
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/pass*.json")):
    for k, v in json.load(open(f)).items():
        if sys.argv[3] not in k:
            continue
        o = out.setdefault(k, {})
        for c, x in v.items():
            o[c] = round(x["per_dispatch"])
            o.setdefault("avg_us_under_pmc", round(x["avg_us"], 1))
for k, o in out.items():
    d = o.setdefault("derived", {})
    if o.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_over_idx_active"] = round(o.get("SQ_LDS_BANK_CONFLICT", 0) / o["SQ_LDS_IDX_ACTIVE"], 4)
    if o.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in o:
        d["mfma_busy_over_sq_busy_x_simds(note: counters are summed over their instances)"] = round(o["SQ_VALU_MFMA_BUSY_CYCLES"] / o["SQ_BUSY_CYCLES"], 3)
    if o.get("SQ_WAVE_CYCLES"):
        d["waves_issuing_frac"] = round(o.get("SQ_ACTIVE_INST_ANY", 0) / o["SQ_WAVE_CYCLES"], 3)
        d["waves_issue_stalled_frac"] = round(o.get("SQ_WAIT_INST_ANY", 0) / o["SQ_WAVE_CYCLES"], 3)
    if "FETCH_SIZE" in o:
        d["fabric_read_bytes(FETCH_SIZE KiB x2, gfx950 note)"] = o["FETCH_SIZE"] * 2048
json.dump({"source": f"rocprofv3 --kernel-trace --pmc <group> -- python {sys.argv[4]} (one pass per counter group: tools/gpu_lease.sh); counter summed over its instances per dispatch, averaged over the dispatches (tools/rocpd_pmc.py)",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, o in out.items():
    print(k[:70], o.get("avg_us_under_pmc"), o["derived"])
PY
      ;;
    pmc_gemv)
      mkdir -p gpurun_out/pmc_${RT}
      for v in bf16 fp8; do for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_$c
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python $R/tools/probe_r2.py gemv $v > $R/gpurun_out/pmc_${RT}/gemv_${v}_$c.log 2>&1)
        python tools/rocpd_pmc.py "$(find /tmp/pmc_$c -name '*.db' | head -1)" gpurun_out/pmc_${RT}/gemv_${v}_$c.json > /dev/null 2>> gpurun_out/pmc_${RT}/gemv_${v}_$c.log
      done
      python tools/gemv_traffic.py gpurun_out/pmc_${RT}/gemv_${v}_FETCH_SIZE.json gpurun_out/pmc_${RT}/gemv_${v}_WRITE_SIZE.json gpurun_out/${RT}_pmc_gemv_traffic_$v.json $v | tail -6
      done ;;
    bench)
      timeout 1500 python bench.py > gpurun_out/${RT}_bench_default.log 2> gpurun_out/${RT}_bench_default.err; echo "bench exit $?"
      tail -n 1 gpurun_out/${RT}_bench_default.log > gpurun_out/${RT}_bench_default_line.json
      line gpurun_out/${RT}_bench_default.log "default:" ;;
    configs)
      : > gpurun_out/${RT}_other_configs.txt
      for cfg in "--pair" "--fp8" "--fp8 --batch 64" "--fp8 --batch 128" "--fp8-mx --tile 896 --batch 16 --static" "--no-overlap" "--drain" "--batch 64" "--batch 128" "--batch 1"; do
        timeout 900 python bench.py $cfg --steps 2 --warmup 1 $QUIET > gpurun_out/${RT}_cfg.log 2> gpurun_out/${RT}_cfg.err
        line gpurun_out/${RT}_cfg.log "bench.py $cfg:" | tee -a gpurun_out/${RT}_other_configs.txt
      done ;;
    *) echo "unknown stage $stage" ;;
  esac
done
