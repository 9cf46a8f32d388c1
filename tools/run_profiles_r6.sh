# Round-6 evidence run on ONE MI355X box, from the tree as shipped: the default bench line (batch sweep, other_configs, parity
# self-check), rocprofv3 kernel stats of the same command (eager launches: rocprofv3 segfaults inside hipGraph capture on this image)
# and the three separate PMC passes bench.py's roofline.traffic / attention.mfma_busy_pmc are read from.  Every summary gets the
# fingerprint of the kernel sources it was collected from in its first line; bench.py only replays a summary whose fingerprint is
# this tree's.  A second bench line AFTER the summaries are in place shows the replayed fields next to the live ones.
# usage: bash tools/run_profiles_r6.sh [out-dir-name] [B]
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r6p}; B=${2:-128}; mkdir -p $O
cd $R
FP=$(python -c "import bench; print(bench.csrc_fingerprint())")
export TMPDIR=/tmp
cd /tmp
SDV_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-walk-pass --no-other-configs --no-parity-check > $O/kt_bench.json 2> $O/kt.err; echo "kt rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python $R/tools/unet_once.py $B > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python $R/tools/unet_once.py $B > $O/p2.log 2>&1; echo "p2 rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p3 -- python $R/tools/unet_once.py $B > $O/p3.log 2>&1; echo "p3 rc=$?"
cd $R
python tools/pmc_summary.py $O/round6_pmc_unet_b$B.csv $(find $O/p1 $O/p2 $O/p3 -name "*counter_collection.csv") | tail -2
KS=$(find $O/kt -name "*kernel_stats.csv" | head -1)
(echo "# csrc=$FP rocprofv3 --kernel-trace --stats of: SDV_NO_GRAPH=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-walk-pass --no-other-configs --no-parity-check"; cat $KS) > $O/round6_bench_b${B}_kernel_stats.csv
head -6 $O/round6_bench_b${B}_kernel_stats.csv | cut -c1-200
# the summaries go where bench.py looks for them (this copy of the tree only - commit them from gpurun_out/ afterwards)
cp $O/round6_pmc_unet_b$B.csv $O/round6_bench_b${B}_kernel_stats.csv $R/profiles/
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
du -sh $O
