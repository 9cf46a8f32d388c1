# Round-4 evidence run on one MI355X box: the bench line (with other_configs + the batch-128 parity self-check), rocprofv3 kernel
# stats of the same command (eager launches: rocprofv3 segfaults inside hipGraph capture on this image) and the three separate PMC
# passes bench.py's roofline.traffic / attention.mfma_busy_pmc are read from.  usage: bash tools/run_profiles_r4.sh [out-dir-name]
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r4p}; mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
export TMPDIR=/tmp
cd /tmp
SDV_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-walk-pass --no-other-configs --no-parity-check > $O/kt_bench.json 2> $O/kt.err; echo "kt rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python $R/tools/unet_once.py 128 > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python $R/tools/unet_once.py 128 > $O/p2.log 2>&1; echo "p2 rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p3 -- python $R/tools/unet_once.py 128 > $O/p3.log 2>&1; echo "p3 rc=$?"
cd $R
python tools/pmc_summary.py $O/round4_pmc_unet_b128.csv $(find $O/p1 $O/p2 $O/p3 -name "*counter_collection.csv") | tail -2
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/round4_bench_b128_kernel_stats.csv \;
head -5 $O/round4_bench_b128_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
du -sh $O
