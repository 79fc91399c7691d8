export TMPDIR=/tmp
mkdir -p gpurun_out/util
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_THREAD_CYCLES_VALU\|SQ_ACTIVE_INST_VALU\|SQ_INST_CYCLES_VALU\|SQ_INSTS_VALU_[A-Z0-9_]*\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_INST_CYCLES_SALU\|SQ_ACTIVE_INST_SCA\|SQ_ACTIVE_INST_MISC\|SQ_INSTS_SMEM\|SQ_INSTS_BRANCH\|SQ_INSTS_SENDMSG\|SQ_WAIT_INST_LDS\|SQ_ACTIVE_INST_FLAT\|SQ_INST_LEVEL_LDS\|SQ_INST_LEVEL_VMEM" | sort -u > gpurun_out/util/avail.txt
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --check-scale 0 --ref-scale 0 --pmc 0"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/util/p1 -o pmc -- python3 bench.py $ARGS > gpurun_out/util/b1.json 2> gpurun_out/util/p1.err
python3 tools/summarize_prof.py gpurun_out/util > gpurun_out/util/summary.txt 2>&1
grep "k_tc_" gpurun_out/util/summary.txt | cut -c1-160
cat gpurun_out/util/avail.txt | tr '\n' ' '
find gpurun_out/util -name "*counter_collection.csv" -size +2M -delete
