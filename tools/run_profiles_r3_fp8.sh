# rocprofv3 summaries of BASELINE config 5 with the block-scaled fp8 MFMA (run through gpurun from the repo root)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
SDV_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --dtype fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-walk-pass > $O/kt_bench_fp8.json 2> $O/kt.err; echo "kt rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p3 -- python $R/tools/unet_once.py 128 fp8 > $O/p3.log 2>&1; echo "p3 rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python $R/tools/unet_once.py 128 fp8 > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python $R/tools/unet_once.py 128 fp8 > $O/p2.log 2>&1; echo "p2 rc=$?"
cd $R
python tools/pmc_summary.py $O/round3_pmc_unet_fp8_b128.csv $(find $O/p1 $O/p2 $O/p3 -name "*counter_collection.csv") | tail -2
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/round3_bench_fp8_b128_kernel_stats.csv \;
head -8 $O/round3_bench_fp8_b128_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
du -sh $O
