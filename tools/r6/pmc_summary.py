#!/usr/bin/env python3
"""Per-kernel summary of the four rocprofv3 PMC passes of tools/r6/pmc_configs.sh (every kernel of the command, not only tconv):
launches, time (counter passes run slower than timing runs), matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (SIMD cycles of the launch =
GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), issue / wait split of the wave cycles, fabric traffic per launch (FETCH_SIZE is in KiB and counts wide
coalesced reads at half their size on gfx950: reads = 2 x FETCH_SIZE x 1024, MI355X_MICROARCH.md HBM section), L2 hit rate, LDS bank-conflict
share of the LDS-array cycles.   python tools/r6/pmc_summary.py DIR OUT.json"""
import collections, csv, glob, json, re, sys


def short(name):
    m = re.search(r"tconv_kernel<(.*?)>", name)
    if m:
        return "tconv<" + m.group(1).replace(" ", "") + ">"
    m = re.search(r"(seg_gemm(?:_big)?<[^>]*>|\w+_kernel(?:<[^>]*>)?)", name)
    return (m.group(1) if m else name[:48]).replace(" ", "")


def agg(d):
    f = glob.glob(f"{d}/*counter_collection.csv")
    cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float); seen = set()
    if not f:
        return cnt, n, dur
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); n[k] += 1
            if "Start_Timestamp" in r:
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return cnt, n, dur


root, out_path = sys.argv[1], sys.argv[2]
sq, n, dur = agg(f"{root}/sq")
fe, nf, _ = agg(f"{root}/fetch")
wr, nw, _ = agg(f"{root}/write")
ld, nl, _ = agg(f"{root}/lds")
total = sum(dur.values()) or 1.0
rows = []
rd_all = wt_all = 0.0
for k in sorted(sq, key=lambda k: -dur[k]):
    c = sq[k]; N = n[k]; wc = c["SQ_WAVE_CYCLES"] or 1.0
    ga = c["GRBM_GUI_ACTIVE"] / 8 / N                                  # GPU-active cycles of one launch (summed over the 8 XCDs)
    us = dur[k] / N / 1e3
    rd = 2 * fe[k]["FETCH_SIZE"] * 1024 / max(nf[k], 1); wt = wr[k]["WRITE_SIZE"] * 1024 / max(nw[k], 1)
    hit, miss = wr[k]["TCC_HIT_sum"], wr[k]["TCC_MISS_sum"]
    lds_act = ld[k]["SQ_LDS_IDX_ACTIVE"]
    rows.append(dict(kernel=k, launches=N, share_of_gpu_time=round(dur[k] / total, 4), us_per_launch=round(us, 2), clock_ghz=round(ga / (us * 1e3), 3) if us else None,
                     mfma_busy=round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / N / (ga * 1024), 4) if ga else None,
                     issuing=round(c["SQ_ACTIVE_INST_ANY"] / wc, 3), wait_inst=round(c["SQ_WAIT_INST_ANY"] / wc, 3), wait_any=round(c["SQ_WAIT_ANY"] / wc, 3),
                     read_MB_per_launch=round(rd / 1e6, 3), write_MB_per_launch=round(wt / 1e6, 3), l2_hit=round(hit / (hit + miss), 4) if hit + miss else None,
                     lds_bank_conflict_share=round(ld[k]["SQ_LDS_BANK_CONFLICT"] / lds_act, 4) if lds_act else None,
                     wait_inst_lds=round(ld[k]["SQ_WAIT_INST_LDS"] / (ld[k]["SQ_WAVE_CYCLES"] or 1.0), 4)))
    rd_all += rd * N; wt_all += wt * N
launches = sum(n.values())
calls = 2          # bench.py --steps 1 --warmup 1
out = dict(kernels=rows, launches=launches, calls=calls, fabric_bytes_per_call=(rd_all + wt_all) / calls, fabric_read_bytes_per_call=rd_all / calls,
           fabric_write_bytes_per_call=wt_all / calls,
           note="bench.py --steps 1 --warmup 1: the counters cover warm-up + timed call; per-launch figures are averages over both; separate rocprofv3 --pmc passes "
                "(SQ | FETCH_SIZE | WRITE_SIZE, TCC_HIT, TCC_MISS | LDS); reads = 2 x FETCH_SIZE KiB (gfx950 correction)")
json.dump(out, open(out_path, "w"), indent=1)
print("%-58s %6s %6s %8s %6s %6s %6s %6s %8s %8s %6s %7s" % ("kernel", "n", "share", "us", "GHz", "mfma", "issue", "waitI", "rd MB", "wr MB", "L2hit", "ldsBC"))
for r in rows[:22]:
    print("%-58s %6d %6.3f %8.1f %6s %6s %6.3f %6.3f %8.2f %8.2f %6s %7s" % (r["kernel"][:58], r["launches"], r["share_of_gpu_time"], r["us_per_launch"], r["clock_ghz"], r["mfma_busy"],
                                                                       r["issuing"], r["wait_inst"], r["read_MB_per_launch"], r["write_MB_per_launch"], r["l2_hit"], r["lds_bank_conflict_share"]))
print("fabric bytes per call (the profiled run = %d calls): read %.1f MB, write %.1f MB; %d launches per call" % (calls, rd_all / calls / 1e6, wt_all / calls / 1e6, launches // calls))
