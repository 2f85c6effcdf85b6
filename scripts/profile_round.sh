#!/bin/bash
# Produces the per-round evidence kept under profiles/: kernel-trace stats of the bench command and the HBM-traffic
# PMC passes of the fused kernel (FETCH_SIZE and WRITE_SIZE in separate passes, MI355X_MICROARCH.md "HBM").
# usage (on the GPU box): scripts/profile_round.sh r03   (then, here: scripts/make_profile_summary.py r03)
tag=${1:-r06}
export TMPDIR=/tmp
out=/root/repo/gpurun_out/$tag
rm -rf $out
mkdir -p $out
# plain bench lines first (counter collection leaves the GPU in the profiling power state).  Since round 3 the default schedule is
# the step replayed from a hipGraph at every N; --eager lines ride along for the host-launched step
python3 /root/repo/bench.py > $out/${tag}_bench.json 2>/dev/null
python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_driver_args.json 2>/dev/null     # the driver's command line
for c in C2 C3 C4shard C5 headline+head; do python3 /root/repo/bench.py --config $c > $out/${tag}_bench_$c.json 2>/dev/null; done
# cfg.dg_exact_masks (DG_EXACT_MASKS): the dense ViT-S grids with the clamp masks from fp32-quality cd
python3 /root/repo/bench.py --exact-masks --no-cpu-baseline > $out/${tag}_bench_exact_masks.json 2>/dev/null
python3 /root/repo/bench.py --config C5 --exact-masks --no-cpu-baseline > $out/${tag}_bench_C5_exact_masks.json 2>/dev/null
for c in headline C2 C3 C4shard; do python3 /root/repo/bench.py --config $c --eager --no-cpu-baseline > $out/${tag}_bench_${c}_eager.json 2>/dev/null; done
# the N > 1 schedule with one rank (RCCL initialised, step replayed from two hipGraphs, collective on the side stream)
python3 /root/repo/bench.py --force-dist --no-cpu-baseline 2>/dev/null | grep "^{" > $out/${tag}_bench_force_dist.json
python3 /root/repo/scripts/parity_table.py $out/${tag}_parity.md > /dev/null 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 /root/repo/bench.py --steps 50 --warmup 5 --clock-warmup-s 0.25 --no-cpu-baseline > $out/bench_under_rocprof.json 2>/dev/null )
cp $out/stats/*/*kernel_stats.csv $out/${tag}_kernel_stats.csv
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 /root/repo/bench.py --steps 3 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 /root/repo/bench.py --steps 3 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/pmc_sq -- python3 /root/repo/bench.py --steps 3 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline > /dev/null 2>&1 )
# per-kernel tables of the small-grid configurations (round 5: regenerated every round) and of the step with the head (round 6)
for c in C2 C3 C4shard headline+head; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python3 /root/repo/bench.py --config $c --steps 50 --warmup 5 --clock-warmup-s 0.25 --no-cpu-baseline > /dev/null 2>&1 )
  cp $out/stats_$c/*/*kernel_stats.csv $out/${tag}_kernel_stats_$c.csv
done
cp $out/${tag}_kernel_stats.csv $out/${tag}_kernel_stats_headline.csv
python3 -m pytest /root/repo/tests -m gpu -q > $out/${tag}_gputests.txt 2>&1
python3 - <<PY
import csv, glob, collections, json
res = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = glob.glob("$out/%s/*/*counter_collection.csv" % name)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        if k.startswith(("k_", "void k_")):
            res.setdefault(k, {}).update({c: val / len(disp[k]) for c, val in v.items()})
main = [k for k in res if "k_corr2" in k or "k_corr_main" in k][0]
m = res[main]
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide stream -> x2
m["hbm_read_bytes_corrected"] = 2 * m["FETCH_SIZE"] * 1024
m["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
m["hbm_traffic_bytes_per_launch"] = m["hbm_read_bytes_corrected"] + m["hbm_write_bytes"]
json.dump(res, open("$out/${tag}_pmc_per_launch.json", "w"), indent=1, sort_keys=True)
print(main, json.dumps(m, indent=1))
PY
head -14 $out/${tag}_kernel_stats.csv | cut -c1-150
cat $out/bench_under_rocprof.json
# (the raw traces are tens of MB: only the summaries travel back)
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete
