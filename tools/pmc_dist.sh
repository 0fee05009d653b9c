# SQ/TA counters of the merged distortion kernels (separate --pmc passes, no tracing domains)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_WAVES" "TA_FLAT_READ_WAVEFRONTS TA_TOTAL_WAVEFRONTS TA_BUSY_ SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d gpurun_out/pmc_dist$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>gpurun_out/pmc_dist$i.err || tail -3 gpurun_out/pmc_dist$i.err
done
