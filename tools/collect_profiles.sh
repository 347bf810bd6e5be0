#!/bin/bash
# Collect the round's measurements on the GPU box (run through gpurun from the repo root):
#   bench JSON lines, rocprofv3 kernel traces (--kernel-trace --stats) and the two PMC passes (FETCH_SIZE / WRITE_SIZE) per
#   workload.  The raw rocprof outputs are reduced on the box (tools/rocpd_summary.py, tools/pmc_traffic.py) to the small
#   files that are then copied into profiles/ and committed; only those travel back (gpurun_out/ is capped at 64 MiB).
TAG=${1:-r05}
WHAT=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
mkdir -p "$O"
RAW=/tmp/raw_$TAG
rm -rf "$RAW"; mkdir -p "$RAW"
run() { local n=$1; shift; echo "== $n: $*"; "$@" > "$O/$n.out" 2> "$O/$n.err"; local rc=$?; echo "rc=$rc"; [ $rc -ne 0 ] && tail -5 "$O/$n.err"; return 0; }
prof() { local n=$1; shift
  run "${n}_trace" rocprofv3 --kernel-trace --stats -d "$RAW/${n}_trace" -o t -- "$@"
  local db; db=$(find "$RAW/${n}_trace" -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" "$O/${n}" | tail -1
  # counter passes: eager launches, and the launch-per-op depth phase (the persistent frame kernels need every CU resident at once,
  # which a counter-collecting profiler does not promise; the traffic figures are those of the GEMV / GEMM kernels anyway)
  RST_DEPTH_FRAME=0 NO_CUDA_GRAPH=1 run "${n}_fetch" rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$RAW/${n}_fetch" -o f -- "$@"
  RST_DEPTH_FRAME=0 NO_CUDA_GRAPH=1 run "${n}_write" rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$RAW/${n}_write" -o w -- "$@"
  local fc wc; fc=$(find "$RAW/${n}_fetch" -name "*counter_collection.csv" | head -1); wc=$(find "$RAW/${n}_write" -name "*counter_collection.csv" | head -1)
  [ -n "$fc" ] && [ -n "$wc" ] && python tools/pmc_traffic.py "$fc" "$wc" "$O/${n}_pmc_traffic.json" | head -6
  rm -f "$O/${n}_trace.out" "$O/${n}_fetch.out" "$O/${n}_write.out"
}
# trace only (no counter passes): prof_trace <name> <cmd...>
prof_trace() { local n=$1; shift
  run "${n}_trace" rocprofv3 --kernel-trace --stats -d "$RAW/${n}_trace" -o t -- "$@"
  local db; db=$(find "$RAW/${n}_trace" -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" "$O/${n}" | tail -1
  rm -f "$O/${n}_trace.out"
}
# the tools build of the library (measurement knobs, op-boundary stamps) does not travel with the snapshot (.gpurunignore): the sections
# that need it build it on the box
need_ablation() { [ -f rstnet_amd/librstnet_hip_ablation.so ] || make -C rstnet_amd/csrc -j32 ablation > "$O/ablation_build.log" 2>&1 || tail -5 "$O/ablation_build.log"; }
# the bench lines quote the trace / counter summaries of the SAME code: each workload is profiled first, its summaries are copied
# into this box's profiles/ under the tag, and only then the bench line is taken
publish() { for f in "$O"/$1_*; do case "$f" in *.out|*.err|*_bench.json) ;; *) cp "$f" "profiles/${TAG}_$(basename "$f")";; esac; done
  # every published kernel-trace summary names the build of the library it was taken with (bench.py quotes it for that build only)
  for f in profiles/${TAG}_$1_kernel_stats.csv; do [ -f "$f" ] && python tools/profile_meta.py "${f%.csv}" > /dev/null && cp "${f%.csv}.meta.json" "$O/$1_kernel_stats.meta.json"; done; }
if [ "$WHAT" = all ] || [ "$WHAT" = lm ]; then
  prof lm python bench.py --workload lm --steps 6 --warmup 2 --no-cpu-baseline --timing-samples 2
  db=$(find "$RAW/lm_trace" -name "*.db" | head -1); [ -n "$db" ] && python tools/frame_timeline.py "$db" "$O/lm_timeline.csv" lm_ring_begin_kernel 2
  publish lm
  # batch 32 (configs[2] at the per-GPU stream count): its own trace + counter summaries -> profiles/TAG_lm32_*
  prof lm32 python bench.py --workload lm --lm-batch 32 --steps 6 --warmup 2 --no-cpu-baseline --timing-samples 2
  publish lm32
  run lm_bench python bench.py --workload lm --steps 60 --warmup 5
  run lm32_bench python bench.py --workload lm --lm-batch 32 --steps 60 --warmup 5 --no-cpu-baseline
  # (ring offset 3000, batch 32 and the batch-32 end-to-end frame ride on the default line as sub-objects since round 3)
  BENCH_DEPTH_CHAINS=1 python tools/bench_depth.py 2>&1 | grep -v amdgpu > "$O/depth_phase.txt"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = gpt ]; then
  prof gpt python bench.py --workload gpt --steps 6 --warmup 2 --no-cpu-baseline
  publish gpt
  prof gpt_fp8 python bench.py --workload gpt --fp8 --steps 6 --warmup 2 --no-cpu-baseline
  publish gpt_fp8
  run gpt_bench python bench.py --workload gpt --steps 40 --warmup 5
  run gpt_fp8_bench python bench.py --workload gpt --fp8 --steps 40 --warmup 5 --no-cpu-baseline
  run gpt1_bench python bench.py --workload gpt --lm-batch 1 --steps 60 --warmup 5 --no-cpu-baseline
fi
if [ "$WHAT" = all ] || [ "$WHAT" = e2e ]; then
  run e2e1_trace rocprofv3 --kernel-trace --stats -d "$RAW/e2e1_trace" -o t -- python bench.py --workload e2e --lm-batch 1 --steps 30 --warmup 4 --no-cpu-baseline --timing-samples 2
  db=$(find "$RAW/e2e1_trace" -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py "$db" "$O/e2e1" | tail -1
  [ -n "$db" ] && python tools/frame_timeline.py "$db" "$O/e2e1_timeline.csv" lm_ring_begin_kernel 2
  rm -f "$O/e2e1_trace.out"
  # 32 streams (configs[3] at its per-GPU size): kernel trace -> profiles/TAG_e2e32_kernel_stats.csv
  prof_trace e2e32 python bench.py --workload e2e --lm-batch 32 --steps 20 --warmup 4 --no-cpu-baseline --timing-samples 2
  publish e2e1
  publish e2e32
  run e2e1_bench python bench.py --workload e2e --lm-batch 1 --steps 60 --warmup 6
  run e2e32_bench python bench.py --workload e2e --lm-batch 32 --steps 60 --warmup 6 --no-cpu-baseline
fi
if [ "$WHAT" = all ] || [ "$WHAT" = phases ]; then
  need_ablation
  # op-boundary stamps of the two persistent kernels (tools build: librstnet_hip_ablation.so)
  (python tools/probes/codec_tr_phases.py 1 2; python tools/probes/codec_tr_phases.py 1 1; python tools/probes/codec_tr_phases.py 2 2) 2>&1 | grep -v amdgpu.ids > "$O/codec_tr_phases.txt"
  (python tools/probes/depth_frame_phases.py 1; python tools/probes/depth_frame_phases.py 2) 2>&1 | grep -v amdgpu.ids > "$O/depth_frame_phases.txt"
  (python tools/probes/temporal_frame_phases.py --layers 8 --pos 10 --wg 133; python tools/probes/temporal_frame_phases.py --layers 8 --pos 3100 --wg 133) 2>&1 | grep -v amdgpu.ids > "$O/temporal_frame_phases.txt"
  cp "$O/temporal_frame_phases.txt" "profiles/${TAG}_temporal_frame_phases.txt" 2>/dev/null
  cp "$O/codec_tr_phases.txt" "$O/depth_frame_phases.txt" profiles/ 2>/dev/null; for f in codec_tr_phases depth_frame_phases; do mv "profiles/$f.txt" "profiles/${TAG}_$f.txt"; done
fi
# the codec workload last: its default line carries the LM / GPT / end-to-end sub-objects, which quote the summaries published above
if [ "$WHAT" = all ] || [ "$WHAT" = codec ]; then
  prof codec python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sub --no-check --timing-samples 1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = codec ] || [ "$WHAT" = mfma ]; then
  # MFMA-pipe busy cycles of the codec step (its own PMC pass; tools/pmc_mfma.py)
  NO_CUDA_GRAPH=1 run codec_mfma rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$RAW/codec_mfma" -o m -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sub --no-check --timing-samples 1
  mc=$(find "$RAW/codec_mfma" -name "*counter_collection.csv" | head -1)
  [ -n "$mc" ] && python tools/pmc_mfma.py "$mc" "$O/codec_mfma.json" | head -8
  rm -f "$O/codec_mfma.out"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = codec ] || [ "$WHAT" = b3 ]; then
  # the three-plane GEMM: wave-stall / LDS-conflict counters (tools/pmc_stalls.py) and the ablation table of DESIGN.md 3.1b (tools
  # build of the library, RST_B3_DBG: WRONG results by construction, timings only)
  NO_CUDA_GRAPH=1 run b3_stalls_raw rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$RAW/b3_stalls" -o m -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sub --no-check --timing-samples 1
  python tools/pmc_stalls.py "$RAW/b3_stalls" > "$O/b3_stalls.txt"; head -4 "$O/b3_stalls.txt"
  rm -f "$O/b3_stalls_raw.out"
  need_ablation
  if [ -f rstnet_amd/librstnet_hip_ablation.so ]; then
    : > "$O/b3_ablation.txt"
    for v in "RST_B3_DBG=0 full" "RST_B3_FORCE_MASK=1 masked_instance_on_the_edge-free_launches_too" "RST_B3_DIRW=0 weights_staged_through_LDS_(round_4_form)" "RST_B3_WIDE=0 128-wide_tiles_only" "RST_B3_DBG=5 lds_writes_of_unsplit_bits_(no_split_VALU)" "RST_B3_DBG=1 no_split_no_lds_writes" "RST_B3_DBG=2 no_global_loads" "RST_B3_DBG=3 matrix_instructions_only" "RST_B3_DBG=4 no_barriers"; do
      set -- $v
      env "$1" python tools/ab.py LIB=rstnet_amd/librstnet_hip_ablation.so -- --steps 6 --warmup 2 --no-sub --no-cpu-baseline --no-check --timing-samples 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-44s %-14s step %7.3f ms  three-plane GEMMs %7.3f ms  %7.1f TFLOP/s (fp32-equivalent)' % ('$2', '$1', d['ms_per_step'], r['kernel_ms_per_step'], r['achieved']))" >> "$O/b3_ablation.txt"
    done
    cat "$O/b3_ablation.txt"
  fi
fi
if [ "$WHAT" = all ] || [ "$WHAT" = codec ] || [ "$WHAT" = rb ]; then
  # the fused residual blocks alone (tools/bench_resblock.py): times, wave-stall split, instruction counts per launch
  python tools/bench_resblock.py --iters 10 2>&1 | grep -v amdgpu.ids > "$O/rb_times.txt"
  python tools/bench_resblock.py --iters 10 --f32 2>&1 | grep -v amdgpu.ids >> "$O/rb_times.txt"
  cat "$O/rb_times.txt"
  NO_CUDA_GRAPH=1 run rb_stalls_raw rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$RAW/rb_stalls" -o m -- python tools/bench_resblock.py --iters 2
  python tools/pmc_stalls.py "$RAW/rb_stalls" | grep resblock > "$O/rb_stalls.txt"; cat "$O/rb_stalls.txt"
  rm -f "$O/rb_stalls_raw.out"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = codec ]; then
  publish codec
  publish b3
  publish rb
  run codec_bench python bench.py --steps 20 --warmup 5
fi
# the bench lines of the sections that ran -> profiles/TAG_<workload>_bench.json (what tests/test_bench_contract_cpu.py reads)
for f in "$O"/*_bench.out; do [ -s "$f" ] && grep -q '^{' "$f" && cp "$f" "profiles/${TAG}_$(basename "${f%.out}").json"; done
du -sh "$O"; ls "$O"
