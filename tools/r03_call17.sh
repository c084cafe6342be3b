#!/bin/bash
# where the Python sharded driver's host time goes (one rank): cProfile of bench.py --mode shard
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call17; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--workload', 'starknet_2p20', '--mode', 'shard', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-north-star']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(60)
open('$O/cprofile_shard.txt', 'w').write(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40)
open('$O/cprofile_shard_tottime.txt', 'w').write(s.getvalue())
" > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err; head -c 300 $O/bench.json; echo
grep -v "^$" $O/cprofile_shard_tottime.txt | head -60
echo done
