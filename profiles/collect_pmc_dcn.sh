#!/bin/bash
# Counter evidence for the DCN kernels (round 3): what the units of a CU are doing under each of them.
# Separate rocprofv3 passes (--pmc with --kernel-trace only), one DCN layer forward + backward per shape
# (profiles/dcn_layer.py).  usage (GPU box): bash profiles/collect_pmc_dcn.sh -> gpurun_out/pmc_dcn.{md,json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_dcn
rm -rf $O; mkdir -p $O
pass() {   # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/profiles/dcn_layer.py --offsets small > $O/$n.log 2>&1 || echo "pass $n failed" >> $O/failed.txt
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
pass sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass ta1 TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
pass tcp2 TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass mem1 FETCH_SIZE GRBM_GUI_ACTIVE
pass mem2 WRITE_SIZE GRBM_GUI_ACTIVE
python3 $R/profiles/summarise_pmc_dcn.py $O $R/gpurun_out
