"""profiles/r6_pmc_conv_raw.txt (tools/r6/probe*.sh: separate rocprofv3 --kernel-trace --pmc passes per counter group and conv class,
per-launch averages) -> profiles/r6_pmc_conv.json in the format of profiles/r5_pmc_gemm.json.

Normalisations (checked on the 320->320 @64x64 launch against round 4's file): GRBM_GUI_ACTIVE is summed over the 8 XCDs,
SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs -> matrix-pipe busy fraction = MFMA_BUSY / (128 x GRBM); SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* are quad-cycles summed over waves; FETCH_SIZE (KB) x 2 = the gfx950 wide-read correction of MI355X_MICROARCH.md."""
import ast
import json
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
# class -> (M, Cin, Cout, input pixels, what / launches per forward)
CLASSES = {
    "vae c3 128->128@512": (2097152, 128, 128, 2097152, "VAE / TFA 128 -> 128 @512x512, B=8 (M2097152 N128 K1152: 9 + 1 per forward, 9.0 ms)"),
    "vae c3 256->256@256": (524288, 256, 256, 524288, "VAE 256 -> 256 @256x256, B=8 (M524288 N256 K2304: 8 per forward, 6.2 ms)"),
    "unet c3 1280->1280@16": (2048, 1280, 1280, 2048, "UNet 1280 -> 1280 @16x16, B=8 (M2048 N1280 K11520: 140 per forward)"),
    "unet c3 1280->1280@8": (512, 1280, 1280, 512, "UNet 1280 -> 1280 @8x8, B=8 (M512 N1280 K11520: 220 per forward; whole-image tile measured here, the weight-stream kernel replaced it later in the round)"),
    "s2 c3 640->640@32": (2048, 640, 640, 8192, "UNet downsampler 640 -> 640 stride 2, 32x32 -> 16x16 (20 per forward)"),
    "unet c3 320->320@64": (32768, 320, 320, 32768, "UNet 320 -> 320 @64x64, B=8 (M32768 N320 K2880: 140 per forward, the most frequent conv launch)"),
    "unet c3 1280->1280@16up": (2048, 1280, 1280, 512, "UNet upsampler 1280 -> 1280, 8x8 -> 16x16 fused nearest-2x (20 per forward; generic kernel + reduce at the time of the pass)"),
}


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "r6_pmc_conv_raw.txt"          # (r6_pmc_conv_final_raw.txt: the final tree of the round)
    final = "final" in src
    raw = open(os.path.join(ROOT, "profiles", src)).read().split("== wstream timeline")[0]
    data = {}
    cls = None
    for line in raw.splitlines():
        m = re.match(r"== (.*?) :: ", line)
        if m:
            cls = m.group(1)
            continue
        m = re.match(r"(.*?) grid=(\d+) wg=(\d+) (\{.*\}) launches (\d+)", line)
        if m and cls:
            kern = re.sub(r"\(\(?ano.*$", "", m.group(1)).strip() or "splitk_reduce*_kernel"
            if kern.startswith("ConvK)"):
                kern = "splitk_reduce_gn_kernel (name cut by the 70-character key)"
            elif kern == "ConvK":
                kern = "conv3x3_wstream8_kernel (name cut by the 90-character key)"
            d = data.setdefault(cls, {}).setdefault((kern, int(m.group(2)), int(m.group(3))), {})
            d.update(ast.literal_eval(m.group(4)))
    out = {"round": 6,
           "source": f"profiles/{src} (tools/r6/probe1.sh / probe2.sh, final tree: tools/r6/pmc_conv_final.sh: separate rocprofv3 --kernel-trace --pmc passes per counter group, per-launch "
                     "averages over 18 launches of tools/r6/time_conv.py-style loops; FETCH_SIZE x2 = the gfx950 wide-read correction of MI355X_MICROARCH.md)",
           "normalisation": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE) [GRBM summed over 8 XCDs, MFMA_BUSY over 1024 SIMDs]; "
                            "wave-cycle split in quad-cycles over SQ_WAVE_CYCLES; lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
           "classes": []}
    for cls, kerns in data.items():
        M, cin, cout, inpix, what = CLASSES[cls]
        alg_r = 2 * (inpix * cin + cout * 9 * cin)
        alg_w = 2 * M * cout
        entry = {"class": cls, "what": what, "algorithmic_read_bytes": alg_r, "algorithmic_write_bytes": alg_w,
                 "gflop": round(2.0 * M * cout * 9 * cin / 1e9, 2), "kernels": []}
        for (kern, grid, wg), c in kerns.items():
            k = {"kernel": kern, "workgroups": grid // wg, "threads_per_workgroup": wg}
            if "FETCH_SIZE" in c:
                k["hbm_side_fetch_bytes"] = int(c["FETCH_SIZE"] * 2 * 1024)
            if "WRITE_SIZE" in c:
                k["hbm_side_write_bytes"] = int(c["WRITE_SIZE"] * 1024)
            if "GRBM_GUI_ACTIVE" in c:
                k["gpu_cycles_per_launch"] = round(c["GRBM_GUI_ACTIVE"] / 8)
                k["mfma_busy_frac"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (128.0 * c["GRBM_GUI_ACTIVE"]), 4)
            if "SQ_WAVE_CYCLES" in c:
                w = float(c["SQ_WAVE_CYCLES"])
                k["wave_cycle_split"] = {"waiting_at_waitcnt_or_barrier": round(c.get("SQ_WAIT_ANY", 0) / w, 3),
                                         "issue_stall": round(c["SQ_WAIT_INST_ANY"] / w, 3),
                                         "of_which_lds_issue": round(c["SQ_WAIT_INST_LDS"] / w, 4),
                                         "issuing": round(c["SQ_ACTIVE_INST_ANY"] / w, 3)}
            if c.get("SQ_LDS_IDX_ACTIVE"):
                k["lds_bank_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
            k["raw"] = c
            entry["kernels"].append(k)
        # class-level traffic: sum over the kernels of one launch sequence (conv + its reduce pass); the generic-kernel A/B twin that
        # some passes also recorded for the 16x16 class is left out of the sum
        main_k = [k for k in entry["kernels"] if "hbm_side_fetch_bytes" in k and not (not final and cls == "unet c3 1280->1280@16" and k["kernel"].startswith("igemm_kernel"))]
        # the 16x16 class recorded two reduce launches per conv (one per variant): halve
        f = sum(k["hbm_side_fetch_bytes"] * (0.5 if not final and cls == "unet c3 1280->1280@16" and "reduce" in k["kernel"] else 1) for k in main_k)
        wr = sum(k["hbm_side_write_bytes"] * (0.5 if not final and cls == "unet c3 1280->1280@16" and "reduce" in k["kernel"] else 1) for k in main_k)
        entry["hbm_bytes_per_launch"] = int(f + wr)
        entry["hbm_over_algorithmic"] = round((f + wr) / (alg_r + alg_w), 2)
        out["classes"].append(entry)
    if final:
        out["note"] = ("final tree of round 6: two loader waves in the 8 x 32 halo conv, the wave-specialised whole-image kernel with the weight-major XCD map, the "
                       "8 x 8 weight stream; the '@16' class averages the plain and the fused-upsample launch (the shape filter matches both)")
    dst = os.path.join(ROOT, "profiles", "r6_pmc_conv_final.json" if final else "r6_pmc_conv.json")
    json.dump(out, open(dst, "w"), indent=1)
    for e in out["classes"]:
        ks = e["kernels"][0]
        print(f"{e['class']:28s} HBM {e['hbm_bytes_per_launch'] / 1e6:8.1f} MB = {e['hbm_over_algorithmic']:.2f} x algorithmic; "
              f"{ks['kernel'][:40]:40s} MFMA busy {ks.get('mfma_busy_frac')}, split {ks.get('wave_cycle_split')}, bank conflicts {ks.get('lds_bank_conflict_frac')}")


if __name__ == "__main__":
    sys.exit(main())
