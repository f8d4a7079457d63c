"""Fold the three rocprofv3 passes of tools/collect_profiles.sh into one JSON (profiles/<tag>_kernels.json).

Per kernel family (the labels bench.py's roofline leg uses): launches per step, average launch duration from the
--kernel-trace --stats pass, and HBM bytes per launch from the two PMC passes, corrected as
/opt/skills/guides/MI355X_MICROARCH.md (HBM) and cdna_hip_programming.md §7 prescribe for gfx950:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024        (FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE counts a 128-B
                                                              streaming request as 64 B, WRITE_SIZE is taken as is)
usage: python tools/summarize_profiles.py gpurun_out/prof_r01 512
"""
import collections
import csv
import glob
import json
import re
import sys

FAMILIES = [                      # label (frostnet_amd/engine.py prof tags) <- kernel-name regex; first match wins
    ("stem_fwd_stats", r"k_pw<0, 8, true, true, 0, 2>"), ("stem_fwd_emit", r"k_pw<1, 8, true, true, 0, 2>"),        # SP = 2 (32 channels, 40-byte rows) is the stem's instance only
    ("sq_fwd", r"k_sq_fwd"), ("pw_fwd_stats", r"k_pw<0,|k_dgrad_wide<2,"), ("pw_fwd_emit_add", r"k_pw_ew_emit_add"), ("pw_fwd_emit", r"k_pw<1,|k_pw_ew<2>|k_pwc<2,|k_sq_emit_cat"),
    ("pw_bwd_reduce", r"k_pw<2,|k_pw_ew<0>|k_dgrad_wide<1,|k_pwc<0,"),
    ("pw_bwd_fused", r"k_pw<3, \d+, \w+, \w+, [1-9]\d*[,>]"), ("pw_bwd_dc", r"k_pw<3,|k_pw_ew<1>|k_pwc<1,"), ("pw_dgrad", r"k_pw<4,|k_dgrad_wide<0,"), ("pw_wgrad", r"k_pw_wgrad"),
    ("blk_expand_dw", r"k_blk_expand_dw"), ("blk_dw_reduce", r"k_blk_dw_reduce"), ("blk_dw_stats", r"k_blk_dw_stats"), ("blk_dw_bwd", r"k_blk_dw_bwd"), ("blk_dw_bred", r"k_blk_dw_bred"),          # block-level fused forward kernels (csrc/frost_block.hip)
    ("dw_bwd_one", r"k_dwb_s[12]<"),          # one-sweep depthwise backward (csrc/frost_dwb.hip): dc pass + weight gradient + data gradient (+ conv1's reduce pass)
    ("dw_fwd_stats", r"k_dw3<DwGeo<[^>]*>, 0[,>]|k_dwm<\d, \d+, 0>|k_dws<\d, \d, \d+, \d, 0>"), ("dw_fwd_emit", r"k_dw3<DwGeo<[^>]*>, 1[,>]|k_dwm<\d, \d+, [14]>|k_dws<\d, \d, \d+, \d, 1>"),
    ("dw_bwd_reduce", r"k_dw3<DwGeo<[^>]*>, 2[,>]|k_dws<\d, \d, \d+, \d, 2>"), ("dw_bwd_dc", r"k_dw3<DwGeo<[^>]*>, [34][,>]"),
    ("dw_wgrad", r"k_dw3_wgrad"), ("dw_dgrad", r"k_dw3_dgrad"),
    ("conv_finalize", r"k_conv_finalize"), ("wgrad_finalize", r"k_wgrad_finalize|k_weight_grad_finalize"),
    ("cat_fwd", r"k_cat_requant|k_cat_observe"), ("cat_bwd", r"k_cat_bwd"), ("add_fwd_minmax", r"k_add_minmax"),
    ("add_fwd_emit", r"k_add_requant"), ("add_bwd", r"k_add_bwd"), ("stem_im2col", r"k_stem_im2col|k_stem_wgrad_remap"),
    ("gradboost", r"k_gradboost"), ("weight_prep", r"k_wprep|k_save_sigma|k_stats_init"),
    ("head", r"k_avgpool|k_classifier|k_head|k_sgemm|k_mask_logits|k_softmax_ce|k_dropout|k_fake_quant|k_minmax|k_fill_minmax|k_observer_update|k_quantize_input|k_pool"),
    ("torch_elementwise", r"at::native|elementwise_kernel|vectorized_elementwise|__amd_rocclr_(fill|copy)Buffer|reduce_kernel"),
]


def family(name):
    for lab, rx in FAMILIES:
        if re.search(rx, name):
            return lab
    return None


def main():
    root, batch = sys.argv[1], int(sys.argv[2])
    out = collections.defaultdict(dict)
    f = glob.glob(f"{root}/stats/**/*kernel_stats.csv", recursive=True)
    total_ns = 0.0
    if f:
        agg = collections.defaultdict(lambda: [0, 0.0])
        other = [0, 0.0]
        for r in csv.DictReader(open(f[0])):
            lab = family(r["Name"])
            tgt = agg[lab] if lab else other
            tgt[0] += int(r["Calls"]); tgt[1] += float(r["TotalDurationNs"])
            total_ns += float(r["TotalDurationNs"])
        for lab, (calls, ns) in agg.items():
            out[lab].update(calls=calls, avg_launch_us=round(ns / calls / 1e3, 2), total_ms=round(ns / 1e6, 3))
        out["_other"] = dict(calls=other[0], total_ms=round(other[1] / 1e6, 3))
    for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)
        if not f:
            continue
        per = collections.defaultdict(lambda: [set(), 0.0])
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] != ctr:
                continue
            lab = family(r["Kernel_Name"])
            if lab:
                per[lab][0].add(r["Dispatch_Id"]); per[lab][1] += float(r["Counter_Value"])
        steps = max(1, len(per["gradboost"][0])) if "gradboost" in per else 1          # one optimizer launch per step of the pass
        for lab, (ids, kib) in per.items():
            out[lab][f"{ctr}_KiB_per_launch"] = round(kib / len(ids), 1)
            out[lab][f"{ctr}_launches"] = len(ids)
            out[lab][f"{ctr}_KiB_per_step"] = round(kib / steps, 1)
            out[lab]["launches_per_step"] = round(len(ids) / steps, 2)
    # MFMA / VALU counters (separate --pmc passes): per family totals per launch
    for d_ in sorted(glob.glob(f"{root}/mfma_*")):
        f = glob.glob(f"{d_}/**/*counter_collection.csv", recursive=True)
        if not f:
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(lambda: [set(), 0.0]))
        for r in csv.DictReader(open(f[0])):
            lab = family(r["Kernel_Name"])
            if lab:
                e = per[lab][r["Counter_Name"]]
                e[0].add(r["Dispatch_Id"]); e[1] += float(r["Counter_Value"])
        for lab, cs in per.items():
            for ctr, (ids, val) in cs.items():
                out[lab][f"{ctr}_per_launch"] = round(val / max(1, len(ids)), 1)
    # FETCH_SIZE / WRITE_SIZE calibration (tools/probe_fetch.hip)
    cal = {}
    for sub, ctr in (("cal_fetch", "FETCH_SIZE"), ("cal_write", "WRITE_SIZE")):
        f = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)
        if not f:
            continue
        per = collections.defaultdict(lambda: [set(), 0.0])
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == ctr:
                k = r["Kernel_Name"].split("(")[0]
                per[k][0].add(r["Dispatch_Id"]); per[k][1] += float(r["Counter_Value"])
        for k, (ids, kib) in per.items():
            cal.setdefault(k, {})[f"{ctr}_KiB_per_launch"] = round(kib / len(ids), 1)
    for lab, d in out.items():
        if "FETCH_SIZE_KiB_per_launch" in d and "WRITE_SIZE_KiB_per_launch" in d:
            d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE_KiB_per_launch"] + d["WRITE_SIZE_KiB_per_launch"]) * 1024)
            d["hbm_bytes_per_step"] = int((2 * d["FETCH_SIZE_KiB_per_step"] + d["WRITE_SIZE_KiB_per_step"]) * 1024)
    hbm_step = sum(d.get("hbm_bytes_per_step", 0) for d in out.values())
    launches_step = sum(d.get("launches_per_step", 0) for lab, d in out.items() if lab != "torch_elementwise")
    doc = dict(batch=batch, fetch_calibration=cal, correction="hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950; separate --pmc passes, eager step)",
               stats_total_ms=round(total_ns / 1e6, 3), hbm_bytes_per_step=hbm_step, kernel_launches_per_step=round(launches_step, 1),
               algorithmic_bytes_per_step=45593016 * batch, traffic_ratio=round(hbm_step / (45593016.0 * batch), 3), families=out)
    json.dump(doc, open(f"{root}/summary.json", "w"), indent=1, sort_keys=True)
    for lab, d in sorted(out.items(), key=lambda kv: -kv[1].get("total_ms", 0)):
        print(f"{lab:16s} {d}")


if __name__ == "__main__":
    main()
