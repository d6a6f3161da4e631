#!/bin/bash
# Counter passes over bench.py itself (one pass per counter set: the hardware cannot hold them together).
# Raw rocprofv3 output stays in /tmp on the GPU box; only the per-kernel summaries go to gpurun_out/.
#   bash tools/run_pmc_bench.sh <out dir under gpurun_out> [pass ...]      passes: small fetch write sq
set -u
# PMC_LAYERS=N runs bench.py with --layers N: per-launch counters of a kernel do not depend on the layer count, and
# the full 28-layer process trips a read past the end of a 1-MiB runtime buffer inside librocprofiler-sdk's
# queue-intercept path (profiles/README.md), at a point that depends on the launch history.
LAYERS=${PMC_LAYERS:+--layers $PMC_LAYERS}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1
shift
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
run_pass() {
  local name=$1 extra=$2
  shift 2
  rm -rf /tmp/pmc_$name
  timeout -s KILL 500 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -- \
    python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $extra > $out/pmc_$name.log 2>&1
  local rc=$?
  echo "rc $rc" >> $out/pmc_$name.log
  if [ $rc -ne 0 ]; then exit $rc; fi
  python3 $R/tools/pmc_bench_summary.py /tmp/pmc_$name $out/pmc_$name.json "$name: $*" > $out/pmc_$name.txt 2>&1
  rm -rf /tmp/pmc_$name
}
for p in "$@"; do
  case $p in
    small) run_pass small "--layers 2" SQ_WAVES ;;
    fetch) run_pass fetch "$LAYERS" FETCH_SIZE ;;
    write) run_pass write "$LAYERS" WRITE_SIZE ;;
    sq) run_pass sq "$LAYERS" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE ;;
  esac
done
