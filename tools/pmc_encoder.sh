# PMC passes for the weight-stationary question encoder (each counter group in a run of its own; no tracing beside them).
# usage: bash tools/pmc_encoder.sh <tag>      output: gpurun_out/pmc_<tag>/summary.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
pass() {
  name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/encoder_one.py > $OUT/$name.log 2>&1
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 $R/tools/pmc_summary.py $OUT k_lstm_resident > $OUT/summary.txt
cat $OUT/summary.txt
