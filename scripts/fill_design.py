"""DESIGN.md = docs/DESIGN.in.md with its @@name@@ placeholders filled from the round's bench records under
profiles/ (so that every such number of the document is a number of a committed record; edit the .in file, not
DESIGN.md). usage: python scripts/fill_design.py [rNN]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"


def rec(name):
    p = os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (TAG, name))
    return json.loads(open(p).read().strip().splitlines()[-1])


def f(v, nd):
    return ("%." + str(nd) + "f") % v


def main():
    v = {}
    h = rec("n1")
    rl = h["roofline"]
    v.update(head_value=f(h["value"], 1), head_ms=f(h["ms_per_step"], 2), head_frac=f(rl["frac"], 3),
             head_kernel_ms=f(rl["kernel_ms"], 2), head_kernel_rocprof=f(rl["kernel_ms_rocprof_live"]["kernel_ms"], 2),
             head_frac_rocprof=f(rl["frac_rocprof"], 3), head_busy=f(rl["pipe"]["mfma_busy"], 3),
             head_ghz=f(rl["pipe"]["ghz"], 2), head_traffic_mb=f(rl["traffic"] / 1e6, 0),
             head_busy_ms=f(h["profile"]["busy_ms_per_step"], 2),
             head_busy_ms_af=f(h["profile"]["busy_ms_per_step_after_first"], 2),
             head_launches=f(h["profile"]["launches_per_step"], 1),
             head_bytes_mb=f(h["hbm"]["bytes_counter"] / 1e6, 0), head_bytes_x=f(h["hbm"]["bytes_counter"] / 112.1e6, 1),
             head_gbps=f(h["hbm"]["gbps_counter"], 0), head_l1="%.1e" % h["parity"]["depth_l1"],
             cpu_cores=str(h["cpu_baseline"]["cores"]), cpu_value=f(h["cpu_baseline"]["value"], 3),
             cpu_ratio="%d" % round(h["value"] / h["cpu_baseline"]["value"], -1))
    s = rec("selected")
    v.update(sel_value=f(s["value"], 0), sel_ms=f(s["ms_per_step"], 2), sel_frac=f(s["roofline"]["frac"], 3))
    for key, name in (("rag", "pairs_ragged"), ("scene", "pairs_scene"), ("n1", "pairs_n1")):
        r = rec(name)
        v[key + "_points"] = "{:,}".format(r["config"]["pairs"]["points"])
        v[key + "_kernel_ms"] = f(r["roofline"]["kernel_ms"], 3 if key != "rag" else 2)
        v[key + "_frac"] = f(r["roofline"]["frac"], 3)
        v[key + "_ms"] = f(r["ms_per_step"], 2)
        v[key + "_value"] = f(r["value"], 1)
    v["scene_frac_old"] = "0.787"   # (scripts/ab_tail_split.sh: whole tiles, same session — profiles/r05_ab_tail_split.txt)
    e = rec("embed")
    v.update(embed_ms=f(e["ms_per_step"], 3), embed_gbps="{:,.0f}".format(e["roofline"]["achieved"]),
             embed_frac=f(e["roofline"]["frac"], 2))
    d = rec("decoders")
    v.update(dec_ms=f(d["ms_per_step"], 2), dec_value=f(d["value"], 0), dec_frac=f(d["roofline"]["frac"], 2))
    c3 = rec("config3")
    v.update(ief_ms=f(c3["roofline_stage2"]["ief_rows"]["kernel_ms"], 4), ief_frac=f(c3["roofline_stage2"]["ief_rows"]["frac"], 3),
             c3_ms=f(c3["ms_per_step"], 2))
    hh = rec("f16x3")
    v.update(h_value=f(hh["value"], 0), h_frac=f(hh["roofline"]["frac"], 3))
    tq, tr, tn = rec("train-query"), rec("train-refine"), rec("train")
    v.update(tq_ms=f(tq["ms_per_step"], 2), tq_frac=f(tq["roofline"]["frac"], 2), train_ms=f(tn["ms_per_step"], 2),
             tr_ms=f(tr["ms_per_step"], 2), tr_frac=f(tr["roofline"]["frac"], 2),
             tr_launches="%d" % round(tr["profile"]["launches_per_step"]))
    c2, c4 = rec("config2"), rec("config4")
    v.update(c2_value=f(c2["value"], 1), c2_frac=f(c2["roofline"]["frac"], 3), c4_value=f(c4["value"], 1),
             c4_frac=f(c4["roofline"]["frac"], 3))
    e1, e1o, e4, e3 = rec("e2e_frame_f1"), rec("e2e_frame_f1_onestream"), rec("e2e_frame_f4"), rec("e2e_frame_f1_streams3")
    v.update(e2e_f1_ms=f(e1["ms_per_frame"], 3), e2e_one_ms=f(e1o["ms_per_frame"], 3), e2e_f4_ms=f(e4["ms_per_frame"], 3),
             e2e_s3_ms=f(e3["ms_per_frame"], 3))
    v.update(e2e_s6_ms=f(rec("e2e_frame_f1_streams6")["ms_per_frame"], 3),
             e2e_sel6_ms=f(rec("e2e_frame_f1_selected_streams6")["ms_per_frame"], 3),
             tq_sel_ms=f(rec("train-query_selected")["ms_per_step"], 2), tq_dense_ms=f(rec("train-query_dense")["ms_per_step"], 2))
    g128, g32 = rec("gf128"), rec("gf32")
    tp = rec("train_pair")
    v.update(train_pair_ms=f(tp["ms_per_step"], 2), train_pair_frac=f(tp["roofline"]["frac"], 2),
             gf32_frac=f(g32["roofline"]["frac"], 2),
             gf32_chain_frac=f((g32["roofline"].get("chain_kernel") or {}).get("frac_rocprof") or 0.0, 2),
             gf128_chain_frac=f((g128["roofline"].get("chain_kernel") or {}).get("frac_rocprof") or 0.0, 2),
             gf32_layers_value=f(rec("gf32_layers")["value"], 0), gf128_layers_value=f(rec("gf128_layers")["value"], 1))
    v.update(gf128_value=f(g128["value"], 1), gf128_frac=f(g128["roofline"]["frac"], 2), gf32_value=f(g32["value"], 0),
             e2e_f8_ms=f(rec("e2e_frame_f8")["ms_per_frame"], 3), e2e_f16_ms=f(rec("e2e_frame_f16")["ms_per_frame"], 3),
             e2e_f16sel_ms=f(rec("e2e_frame_f16_selected")["ms_per_frame"], 3), train_frac=f(tn["roofline"]["frac"], 2))
    rk = e1o.get("roofline_kernels") or {}
    ie, pt = rk.get("ief") or {}, rk.get("points") or {}
    v["ief_frame_us"] = f((ie.get("kernel_ms") or ie.get("kernel_ms_rocprof") or 0.0) * 1e3, 0)
    v.update(e2e_pts_frac=f(pt.get("frac") or 0.0, 3), e2e_pts_frac_rp=f(pt.get("frac_rocprof") or 0.0, 3),
             e2e_ief_frac=f(ie.get("frac") or 0.0, 3), e2e_ief_frac_rp=f(ie.get("frac_rocprof") or 0.0, 3),
             e2e_rccl_ms=f(rec("e2e_rccl_n1")["ms_per_frame"], 3), e2e_graph_ms=f(rec("e2e_graph_f1")["ms_per_frame"], 3),
             e2e_launches=f((e1o.get("profile") or {}).get("launches_per_step_steady") or 0.0, 0))
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(os.path.join(ROOT, "docs", "DESIGN.in.md")).read()
    missing = sorted(set(re.findall(r"@@(\w+)@@", s)) - set(v))
    if missing:
        raise SystemExit("no value for: %s" % missing)
    s = re.sub(r"@@(\w+)@@", lambda m: v[m.group(1)], s)
    open(p, "w").write(s)
    print("filled %d values" % len(v))


if __name__ == "__main__":
    main()
