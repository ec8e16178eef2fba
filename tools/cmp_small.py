import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"])
        for k in d["kernels"]:
            if k["name"].startswith("gemm_sh_64") or k["name"] in ("layernorm","attention_vself"): print("   ",k["name"],k["launches"],k["total_ms"],k["avg_us"])
    except Exception as e: print(f,"fail",e)
