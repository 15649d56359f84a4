#!/bin/bash
# round 5: counters of the bf16 conv kernel and of the bf16 GroupNorm backward; output gpurun_out/pmc_r5_*.txt
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$(pwd)
rm -f gpurun_out/pmc_r5_*.txt
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE"
C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_VALU_MFMA_COEXEC_CYCLES"
D="SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES_EQ_64 SQ_ITEMS"
for shape in "32 128 256" "32 256 128" "32 512 64"; do
  tag=r5_conv_$(echo $shape | tr ' ' '_')
  bash tools/pmc_run.sh $tag "$A" "$B" -- python3 $R/tools/bf16_probe.py conv $shape 3
  bash tools/pmc_run.sh ${tag}_b "$C" "" -- python3 $R/tools/bf16_probe.py conv $shape 3 || true
done
bash tools/pmc_run.sh r5_gnb "$A" "$B" -- python3 $R/tools/gn_probe.py bf16 32 128 256 3
bash tools/pmc_run.sh r5_gnb_b "$C" "" -- python3 $R/tools/gn_probe.py bf16 32 128 256 3 || true
for shape in "32 128 256" "32 256 128" "32 512 64"; do python3 $R/tools/bf16_probe.py conv $shape 30; done > gpurun_out/pmc_r5_timings.txt 2>&1
python3 $R/tools/gn_probe.py bf16 32 128 256 30 >> gpurun_out/pmc_r5_timings.txt 2>&1
