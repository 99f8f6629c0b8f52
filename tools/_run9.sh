cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02h
rm -rf gpurun_out/r02h/pmc3 gpurun_out/r02h/pmc4
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/r02h/pmc3 --output-format csv -- python3 tools/pmc_probe.py encode u64:32 100000000 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM -d gpurun_out/r02h/pmc4 --output-format csv -- python3 tools/pmc_probe.py encode u64:32 100000000 2 > /dev/null 2>&1
echo done
