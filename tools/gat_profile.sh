#!/bin/bash
# rocprofv3 evidence for the GAT half (BASELINE configs[2], N = 4541): kernel trace + MFMA PMC passes.
# usage (on the GPU box): tools/gat_profile.sh TAG ; outputs under gpurun_out/TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02_gat}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/gat_workload.py 4541 200 > $O/unprofiled.log 2>&1
python3 $R/tools/gat_workload.py 1024 200 >> $O/unprofiled.log 2>&1
cat $O/unprofiled.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/gat_workload.py 4541 50 > $O/trace.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1024 -- python3 $R/tools/gat_workload.py 1024 50 > $O/trace1024.log 2>&1 || exit 1
# PMC passes (counters only; SQ has 8 slots, GRBM 2)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- python3 $R/tools/gat_workload.py 4541 20 > $O/pmc1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc2 -- python3 $R/tools/gat_workload.py 4541 20 > $O/pmc2.log 2>&1 || exit 1
ls $O/pmc1/*/ | head
