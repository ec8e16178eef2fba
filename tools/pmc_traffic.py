#!/usr/bin/env python
"""Turn the two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.

    pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [workload]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md, section HBM: FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports one half of the bytes of wide (16 B/lane) coalesced reads -> doubled here;
WRITE_SIZE is uncalibrated and taken as reported.  Kernel names are mapped to the names of the library's HIP-event
recorder (bench.py "kernels"), so bench.py can attach `traffic` to its roofline objects.  Means are over ALL
dispatches of a kernel in the profiled command (the same population bench.py's per-launch averages use).
"""
import collections
import csv
import json
import re
import sys

NAME_MAP = [
    # round 3: the persistent ping-pong kernels (gemm_pp.hip).  Template arguments <EPI, DBG, TAG>: the epilogue code names the
    # Linear (42 mlp.fc1, 32 to_q / to_kv, 40 corr_mlp.fc2, 41 corr_mlp.fc1, 16 input_transform, 36 to_out / mlp.fc2), TAG = 1
    # marks K > 768 (mlp.fc2 against to_out, corr_mlp.fc1, input_transform).  to_q and to_out share one recorder row in bench.py.
    (r"gemm_pp256_kernel<42,", "gemm_sh_pp256_k384_n1536"),
    (r"gemm_pp256_kernel<32,", "gemm_sh_pp256_k384_n768"),
    (r"gemm_pp256_kernel<40,", "gemm_sh_pp256_k384_n256"),
    (r"gemm_pp192_kernel<41,", "gemm_sh_pp192_k2432_n384"),
    (r"gemm_pp192_kernel<16,", "gemm_sh_pp192_k1120_n384"),
    (r"gemm_pp192_kernel<36, false, 1>", "gemm_sh_pp192_k1536_n384"),  # (TAG = 1: K > 768; round 6 removed the fourth template argument)
    (r"gemm_pp192_kernel<36,", "gemm_sh_pp192_k384_n384"),
    (r"gemm_pp192_kernel<32,", "gemm_sh_pp192_k384_n384"),
    (r"gemm_sh_deep64_kernel", "gemm_sh_64x64"),
    (r"conv_pp128_kernel", "conv_pp128"),
    (r"enc_inorm_partial_kernel", "enc_inorm_stats"),
    (r"enc_inorm_apply_kernel", "enc_inorm_apply"),
    (r"enc_stem_im2col_kernel", "enc_stem_im2col"),
    (r"enc_fuse_kernel", "enc_fuse"),
    # the compile-time epilogue code (last template argument) identifies the Linear where it is unique, which lets the
    # per-shape recorder rows of bench.py (gemm_sh_<tile>_k<K>_n<N>) pick up their own traffic:
    #   256x256: 42 = mlp.fc1 (tanh-GELU, SH out), 32 = to_kv (bias only), 40 = corr_mlp.fc2 (SH out)
    #   128x128: 41 = corr_mlp.fc1 (erf-GELU, SH out), 16 = input_transform (bias rows), 32 = to_q; 36 = to_out AND mlp.fc2
    (r"gemm_sh_kernel<2, 4, 4, 2, 2, 42>", "gemm_sh_256_k384_n1536"),
    (r"gemm_sh_kernel<2, 4, 4, 2, 2, 32>", "gemm_sh_256_k384_n768"),
    (r"gemm_sh_kernel<2, 4, 4, 2, 2, 40>", "gemm_sh_256_k384_n256"),
    (r"gemm_sh_kernel<2, 2, 2, 2, 2, 41>", "gemm_sh_128_k2432_n384"),
    (r"gemm_sh_kernel<2, 2, 2, 2, 2, 16>", "gemm_sh_128_k1120_n384"),
    (r"gemm_sh_kernel<2, 2, 2, 2, 2, 32>", "gemm_sh_128_to_q"),
    (r"gemm_sh_kernel<2, 2, 2, 2, 2, 36>", "gemm_sh_128_to_out_and_fc2"),
    (r"gemm_sh_kernel<2, 4, 2, 3,", "gemm_sh_128x384"),
    (r"gemm_sh_kernel<2, 4, 4, 2,", "gemm_sh_256x256"),
    (r"gemm_sh_kernel<2, 2, 2, 2,", "gemm_sh_128x128"),
    (r"gemm_sh_kernel<4, 2, 2, 2, 3", "gemm_sh_256x128x3"),
    (r"gemm_sh_kernel<4, 2, 2, 2,", "gemm_sh_256x128"),
    (r"gemm_sh_kernel<2, 2, 1, 1,", "gemm_sh_64x64"),
    (r"gemm_f16x3_kernel<2, 2>", "gemm_f16x3_128x128"),
    (r"gemm_f16x3_kernel<1, 1>", "gemm_f16x3_64x64"),
    (r"gemm_f32_kernel<2, 2>", "gemm_f32_128x128"),
    (r"gemm_f32_kernel<1, 1>", "gemm_f32_64x64"),
    (r"corr_volume_sh3_kernel", "corr_volume_sh"),  # (round 5: the default sampler)
    (r"corr_volume_sh_kernel", "corr_volume_sh"),
    (r"corr_volume_kernel", "corr_volume"),
    (r"attention_merge_kernel", "attention_merge"),
    (r"attention_time16_kernel", "attention_time"),
    (r"attention_self_kernel", "attention_time"),
    (r"attention_kv64_kernel", "attention_p2v"),   # also the (small) virtual self attention launches
    (r"attention_q64_kernel", "attention_v2p"),
    (r"attention_kernel", "attention_valu"),
    (r"layernorm_kernel", "layernorm"),
    (r"assemble_kernel", "assemble_tokens"),
    (r"heads_kernel", "heads_update"),
    (r"split_rows_scaled_kernel", "pyramid_split"),
    (r"virtual_init_kernel", "virtual_init"),
]


def short(kname):
    for pat, nm in NAME_MAP:
        if pat in kname:
            return nm
    m = re.search(r"(\w+)(<[^>]*>)?\(", kname)
    return (m.group(1) if m else kname)[:48]


def collect(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short(row.get("Kernel_Name", ""))
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
    return acc


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "c3_sliding"  # bench.py workload the passes were collected on
    fetch = collect(fetch_csv, "FETCH_SIZE")
    write = collect(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        fb = 2.0 * 1024.0 * fetch[k][0] / max(fetch[k][1], 1) if k in fetch else None
        wb = 1024.0 * write[k][0] / max(write[k][1], 1) if k in write else None
        res[k] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                  "hbm_bytes_per_launch": (fb or 0.0) + (wb or 0.0),
                  "dispatches": fetch[k][1] if k in fetch else write[k][1],
                  "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH_SIZE x2 (gfx950), KiB -> B"}
    lib_names = {nm for _, nm in NAME_MAP}
    keep = {k: v for k, v in res.items() if k in lib_names}
    keep["_workload"] = workload  # bench.py attaches `traffic` only to this workload's roofline
    # which build of the library the counters were collected on: bench.py refuses to attach traffic measured on another build
    import hashlib
    import os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "co-tracker_amd", "libctk_hip.so")
    keep["_lib_sha256"] = hashlib.sha256(open(so, "rb").read()).hexdigest() if os.path.exists(so) else None
    # ... and the hash of the kernel SOURCES (round 5): the same on every box, so the driver's run recognises the file even if its
    # toolchain produced different bytes; either stamp matching makes the file fresh for bench.py
    import sys as _sys
    _sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as _ge
    keep["_src_sha256"] = _ge.source_hash()
    json.dump(keep, open(out, "w"), indent=1)  # library kernels only (others are printed)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{k:28s} n={v['dispatches']:6d}  fetch {v['fetch_bytes_per_launch'] or 0:14.0f} B  write {v['write_bytes_per_launch'] or 0:14.0f} B")


if __name__ == "__main__":
    main()
