cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VVHIP_WORKLOAD_UNMERGED=1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d gpurun_out/pmc_tu1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d gpurun_out/pmc_tu2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for d in gpurun_out/pmc_tu1 gpurun_out/pmc_tu2; do python tools/rocpd_summary.py $(find $d -name "*.db" | head -1) --pmc | grep -E "tuRdo|hadTile|sadSse|kernel" | cut -c1-60,100-200; done
