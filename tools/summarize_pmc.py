#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in [a for a in sys.argv[1:] if a.startswith("/") or a.startswith("gpurun") or a.startswith(".")]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "at::native" in k or "rocclr" in k:
                continue
            k = k.replace("void ", "").replace("(anonymous namespace)::", "")
            k = re.sub(r"\(.*", "", k)
            out[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in out.items():
    try:
        print(k)
    except BrokenPipeError:
        break
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):.4g}")

# optional: --traffic-json <path> <n_images>  -> bytes per image per kernel from FETCH_SIZE / WRITE_SIZE (KB)
if "--traffic-json" in sys.argv:
    import json
    i = sys.argv.index("--traffic-json")
    path, n_img = sys.argv[i + 1], int(sys.argv[i + 2])
    res, valu = {}, {}
    for k, cs in out.items():
        kk = k.split("<")[0]
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            # gfx950: FETCH_SIZE reports half the bytes of coalesced streaming reads (MI355X_MICROARCH.md, HBM section);
            # calibrated here on fast_tile_kernel, which must read every image byte once: it reports 461 KB per
            # 921.6 KB image -> factor 2.
            f = 2.0 * sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024 / n_img
            w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024 / n_img
            res[kk] = {"fetch": f, "write": w}
        if "SQ_INSTS_VALU" in cs:
            valu[kk] = sum(cs["SQ_INSTS_VALU"]) / len(cs["SQ_INSTS_VALU"]) / n_img
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU (separate passes), bench.py --frames %d" % (n_img // 2),
               "note": "fetch = 2 x FETCH_SIZE (gfx950 under-reports coalesced streaming reads by 2x, MI355X_MICROARCH.md; "
                       "calibrated on fast_tile_kernel whose 921,600-byte image read shows as 461 KB); write = WRITE_SIZE; "
                       "hamming kernels: per image = per pair",
               "bytes_per_image": res,
               "valu_wave_insts_per_image": valu,
               "valu_ns_per_wave_inst_per_simd": {"fast_tile_kernel": 1.7},
               "valu_note": "SQ_INSTS_VALU per image; ns per wave-instruction per SIMD of fast_tile_kernel's instruction mix from "
                            "tools/ubench/valu_rate.hip (profiles/valu_issue_rates_*.txt): v_min3/v_max3_i32, v_alignbyte, v_bfe, "
                            "SDWA and v_dot4 issue at 1.7-1.85 ns, plain add/xor/and at 1.0-1.09 ns"},
              open(path, "w"), indent=1)

# optional: --launch-traffic-json <path>  -> HBM-side bytes per LAUNCH per kernel (BA kernels: one launch = one unit)
if "--launch-traffic-json" in sys.argv:
    import json
    path = sys.argv[sys.argv.index("--launch-traffic-json") + 1]
    res = {}
    for k, cs in out.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            kk = k.split("<")[0]
            f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024
            w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
            res[kk] = {"fetch_raw": f, "fetch": 2.0 * f, "write": w, "launches_sampled": len(cs["FETCH_SIZE"])}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/ba_profile.py at configs[2]",
               "note": "bytes per launch; fetch = 2 x FETCH_SIZE (gfx950 reports half the bytes of wide coalesced reads, "
                       "MI355X_MICROARCH.md; plausibility: chol_syrk must read two 7.3 MB windows + the solved rows once per XCD "
                       "= ~22 MB and write 14.5 MB per launch), write = WRITE_SIZE",
               "bytes_per_launch": res}, open(path, "w"), indent=1)
