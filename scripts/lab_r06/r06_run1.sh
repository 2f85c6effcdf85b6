#!/bin/bash
# round 6, first GPU call: the suite, smoke, today's baselines (headline, exact masks, C2-C4) and the triage of fuzz seed 5133
out=gpurun_out/r06_run1; mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/gputests.txt 2>&1; echo "pytest rc $?" >> $out/gputests.txt
python __graft_entry__.py smoke > $out/smoke.txt 2>&1; echo "smoke rc $?" >> $out/smoke.txt
python bench.py --steps 20 --warmup 5 > $out/bench_headline.json 2> $out/bench_headline.err
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --exact-masks > $out/bench_exact.json 2> $out/bench_exact.err
for c in C2 C3 C4shard; do python bench.py --steps 50 --warmup 5 --no-cpu-baseline --config $c > $out/bench_$c.json 2> $out/bench_$c.err; done
DG_FUZZ_BLOBS=1 python scripts/fuzz_parity.py 1 5133 > $out/fuzz_5133_blobs.txt 2>&1
python scripts/fuzz_parity.py 1 5133 > $out/fuzz_5133_small.txt 2>&1
python scripts/fuzz_parity.py 400 5000 > $out/fuzz_400.txt 2>&1
python scripts/fuzz_parity.py 150 9000 edge > $out/fuzz_edge_150.txt 2>&1
tail -3 $out/gputests.txt; cat $out/smoke.txt | tail -3; cat $out/bench_headline.json | cut -c1-1500; tail -2 $out/fuzz_400.txt
