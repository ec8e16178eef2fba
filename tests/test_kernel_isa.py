"""What hipcc made of the sampler's blend (CPU: hipcc cross-compiles gfx950 without a GPU).

Round 5 tied an intermittent wrong result of sampler version 3 to one compiler-chosen operand form: a packed FP32 op whose LOW
result lane reads the HIGH half of a source (`v_pk_fma_f32 ... op_sel:[0,1,0]`, which hipcc emits when it keeps two weights in one
register pair) right in front of a `ds_write2_b32` of its result: in every build containing the form, lanes 48..63 stored a wrong
first data register; in every build without it, none (profiles/r05_sampler_v3_pk_hazard.txt -- an in-situ correlation: the sequence
replayed in isolation does not fail).  The kernel now hands the packed FMAs plain (w, w) pairs; these tests keep a compiler upgrade
from quietly bringing the form back."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "co-tracker_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_default_sampler_has_no_low_lane_op_sel_packed_ops(tmp_path):
    out = tmp_path / "corr_sh.s"
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
           "--cuda-device-only", os.path.join(CSRC, "corr_sh.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = out.read_text().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*corr_volume_sh3_kernelILi0EE.*:", l)]
    assert len(starts) == 1, "the production instance of corr_volume_sh3_kernel"
    end = next(i for i in range(starts[0], len(lines)) if "s_endpgm" in lines[i])
    body = [l.strip() for l in lines[starts[0]:end]]
    packed = [l for l in body if l.startswith("v_pk_")]
    assert packed, "the blend is expected to use packed FP32 math (if it no longer does, this test can go)"
    bad = [l for l in packed if re.search(r"op_sel:\[", l)]
    assert not bad, f"packed ops whose low lane reads a high half: {bad[:4]}"
    assert sum(l.startswith("v_mfma_f32_16x16x32_f16") for l in body) >= 48


FP_CONTRACT_OFF = {"corr", "corr_sh", "corrblock", "rowops", "v2ops", "encoder", "sampler"}  # as in csrc/Makefile


def _compile_to_asm(name, out_dir):
    out = os.path.join(out_dir, name + ".s")
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    if name in FP_CONTRACT_OFF:
        cmd.append("-ffp-contract=off")
    cmd += ["-S", "--cuda-device-only", os.path.join(CSRC, name + ".hip"), "-o", out]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def _regs(text):
    r = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", text):
        r |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        r.add(int(m.group(1)))
    return r


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_no_kernel_stores_the_result_of_a_low_lane_op_sel_packed_op_right_behind_it(tmp_path):
    """The hazard of round 5 in its general form, over every kernel of the library: a packed FP32 op with a low-lane op_sel
    (`op_sel:[...]` with a 1 in it) whose result register is the DATA of an LDS / global store within the next six
    instructions.  None today (rowops.hip, encoder.hip and corr.hip contain such packed ops, but their results go through
    further VALU instructions first)."""
    from concurrent.futures import ThreadPoolExecutor
    names = sorted(f[:-4] for f in os.listdir(CSRC) if f.endswith(".hip") and f not in ("api.hip", "profile.hip"))
    with ThreadPoolExecutor(max_workers=6) as ex:
        outs = list(ex.map(lambda n: _compile_to_asm(n, str(tmp_path)), names))
    risky = []
    for path in outs:
        lines = [l.strip() for l in open(path).read().split("\n") if l.startswith("\t") and not l.strip().startswith(";")]
        for i, t in enumerate(lines):
            if not (t.startswith("v_pk_") and re.search(r"op_sel:\[[01,]*1", t)):
                continue
            m = re.match(r"\S+\s+v\[(\d+):(\d+)\]", t)
            if not m:
                continue
            dest = set(range(int(m.group(1)), int(m.group(2)) + 1))
            for x in lines[i + 1:i + 7]:
                if re.match(r"(ds_write|ds_store|global_store|buffer_store|flat_store)", x) and (_regs(x) & dest):
                    risky.append((os.path.basename(path), t, x))
                    break
    assert not risky, risky[:5]


def _kernel_body(lines, pattern):
    starts = [i for i, l in enumerate(lines) if re.match(pattern, l)]
    assert len(starts) == 1, (pattern, len(starts))
    end = next(i for i in range(starts[0], len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[starts[0]:end]


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
@pytest.mark.parametrize("tag", [0, 1])
def test_residual_loads_of_the_persistent_gemm_are_untouched_until_their_wait(tmp_path, tag):
    """gemm_pp192_kernel<EPI = 36> ("+ residual": to_out, mlp.fc2) fetches the residual pieces with inline-asm
    global_load_dwordx4 into plain "=v" outputs -- hipcc does not know the values are in flight until the hand-counted vmcnt one
    phase later.  At 255 of 256 VGPRs a compiler-inserted copy, spill or reuse of those registers between the load and the wait
    that retires it would store stale data silently.  This test reads the ISA: no scratch, and between every such load and the first
    s_waitcnt whose vmcnt count retires it (loads retire in order: count <= VMEM operations issued behind it) no instruction
    names its destination registers."""
    asm = _compile_to_asm("gemm_pp", str(tmp_path))
    text = open(asm).read()
    lines = text.split("\n")
    name = f"gemm_pp192_kernelILi36ELb0ELi{tag}E"
    body = [l.strip() for l in _kernel_body(lines, rf"^_ZN.*{name}.*:") if l.startswith("\t") and not l.strip().startswith(";")]
    m = re.search(rf"\.amdhsa_kernel _ZN[^\n]*{name}[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S)
    assert m and re.search(r"\.amdhsa_private_segment_fixed_size 0\b", m.group(1)), "the residual kernel must not use scratch"
    loads = [i for i, t in enumerate(body) if re.match(r"global_load_dwordx4\s+v\[\d+:\d+\],\s*v\[\d+:\d+\],\s*off\b", t)]
    assert len(loads) == 25, len(loads)  # the bias staging loop of the prologue (compiler-visible) + 6 sub-tiles x 4 line pieces per tile
    loads = loads[1:]
    vmem = re.compile(r"(global_load|global_store|buffer_load|buffer_store|flat_load|flat_store|global_atomic)")
    for i in loads:
        d = re.match(r"global_load_dwordx4\s+v\[(\d+):(\d+)\]", body[i])
        dest = set(range(int(d.group(1)), int(d.group(2)) + 1))
        behind = 0
        retired = False
        for t in body[i + 1:i + 4000]:
            w = re.match(r"s_waitcnt\b(.*)", t)
            if w:
                c = re.search(r"vmcnt\((\d+)\)", w.group(1))
                if c and int(c.group(1)) <= behind:
                    retired = True
                    break
                continue
            if vmem.match(t):
                behind += 1
                if re.match(r"global_load_dwordx4\s+v\[", t) and (_regs(t.split(",")[0]) & dest):
                    break  # the destination is re-used by the next sub-tile's load only behind the wait: handled by `retired`
            assert not (_regs(t) & dest), f"{name}: `{t}` touches {sorted(dest)} of `{body[i]}` before the wait that retires it"
        assert retired, f"{name}: no retiring wait found behind `{body[i]}`"
