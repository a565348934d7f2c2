#!/bin/bash
# rocprofv3 kernel trace of the default bench; usage: tools/prof_run.sh <tag> [ENV=val ...]
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
(cd /tmp && env "$@" rocprofv3 --kernel-trace -d $out -o run -- python $OLDPWD/bench.py ${PROF_ARGS:---no-cpu-baseline} > $out/bench.log 2>&1)
db=$(find $out -name "*.db" | head -1)
python tools/rocpd_stats.py $db > $out/all.txt
python tools/rocpd_stats.py $db ${PROF_GY:-256} > $out/batch.txt
tail -1 $out/bench.log | cut -c1-200
cat $out/batch.txt
