#!/bin/bash
# Round-6 hazard experiment: is the sampler's round-5 intermittent wrong result a missing VALU -> LDS wait state?
# Builds corr_sh.hip in the control flow of round 5's deterministic FAILING build (-DCTK_PK_NOP: hipcc emits
# `v_pk_fma_f32 ... op_sel:[0,1,0]` right in front of the `ds_write2_b32` of its result) and inserts K wait states (`s_nop K-1`)
# between the two by a post-pass over the device ISA (any inline asm in the source changes hipcc's choice of the packed form), for
# K = 0 (control: must fail), 1, 2, 3, 4, 8; links each into tools/_ab/libctk_pk<WHERE><K>.so next to the dev objects.
# WHERE=after (default): between the packed op and the LDS write; WHERE=before: in front of the packed op (a VALU -> VALU hazard?).
# On the GPU box: for K in ...; do CTK_LIB_PATH=tools/_ab/libctk_pk${WHERE:-after}$K.so REPS=6 VERS=3 QUIET=1 python tools/soak_corr.py; done
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/co-tracker_amd/csrc
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
make -C $CSRC -j8 dev > /dev/null
mkdir -p $ROOT/tools/_ab
for K in ${KS:-0 1 2 3 4 8}; do
  W=$(mktemp -d)
  cd $W
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DCTK_DEV -DCTK_PK_NOP=$K -c $CSRC/corr_sh.hip -o corr_sh.o -save-temps"
  $HIPCC $FLAGS 2> /dev/null
  $HIPCC $FLAGS -### 2>&1 | grep -E '^ "' > cmds.txt
  S=corr_sh-hip-amdgcn-amd-amdhsa-gfx950.s
  python3 - $S $K ${WHERE:-after} <<'PY'
import re, sys
path, k, where = sys.argv[1], int(sys.argv[2]), sys.argv[3]
before = 0
lines = open(path).read().split("\n")
out, hits, pend = [], 0, None
for l in lines:
    t = l.strip()
    if pend is not None and t and not t.startswith(";") and l.startswith("\t"):
        pend[1] += 1
        if pend[1] > 4:
            pend = None
    if t.startswith("v_pk_") and re.search(r"op_sel:\[[01,]*1", t):
        m = re.match(r"\S+\s+v\[(\d+):(\d+)\]", t)
        if m:
            pend = [{f"v{m.group(1)}", f"v{m.group(2)}"}, 0]
        if where == "before" and k > 0:  # wait states in FRONT of the packed op (between the VALU ops that write its sources and it)
            out.append(f"\ts_nop {k - 1}")
            before += 1
    elif pend is not None and t.startswith("ds_write2_b32") and (set(re.split(r"[ ,]+", t)) & pend[0]):
        hits += 1
        if k > 0 and where == "after":
            out.append(f"\ts_nop {k - 1}")
        pend = None
    out.append(l)
open(path, "w").write("\n".join(out))
print(f"K={k} ({where}): {hits} ds_write2_b32 directly behind a low-lane-op_sel packed op" + (f", s_nop {k - 1} inserted in front of each" if k and where == "after" else "") + (f", s_nop {k - 1} inserted in front of {before} low-lane-op_sel packed ops" if where == "before" else "") + (" (control, untouched)" if not k else ""))
PY
  # re-run: device assembler, lld, bundler; host bitcode (embeds the bundle), host asm, host object
  sed -n '4p;5p;6p;8p;9p;10p' cmds.txt > redo.sh
  bash redo.sh
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_ab/libctk_pk${WHERE:-after}$K.so $(ls $CSRC/dev/*.o | grep -v corr_sh.o) corr_sh.o
  cd $ROOT
  rm -rf $W
done
ls -la $ROOT/tools/_ab/
