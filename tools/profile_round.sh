#!/bin/bash
# End-of-round measurement on the GPU box: bench JSON, rocprofv3 kernel stats of the same
# command (without the extras, so averages are per bench launch), HBM traffic PMC passes,
# per-op table.  usage: bash tools/profile_round.sh <tag>     (writes gpurun_out/<tag>_*)
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
echo "bench done"; tail -c 900 $O/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof -- python3 $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extras --no-1080p > $O/${tag}_prof_bench.json 2> $O/${tag}_prof.err
echo "kernel-trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${tag}_pmc/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-1080p > /dev/null 2> $O/${tag}_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${tag}_pmc/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-1080p > /dev/null 2> $O/${tag}_pmc_write.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${tag}_pmc_hd/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --only-1080p > /dev/null 2> $O/${tag}_pmc_hd_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${tag}_pmc_hd/write -- python3 $R/bench.py --steps 3 --warmup 1 --only-1080p > /dev/null 2> $O/${tag}_pmc_hd_write.err
echo "pmc done"
cd $R
python3 tools/collect_traffic.py $O/${tag}_pmc 128 $O/${tag}_traffic_pmc.json $O/${tag}_pmc_hd
cp $(ls $O/${tag}_prof/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
WARM=20 ITERS=30 python3 tools/bench_ops.py all 64 > $O/${tag}_ops_table.txt 2>&1
echo "ops done"
