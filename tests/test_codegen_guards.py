"""Guards on the GENERATED gfx950 code of the chain kernels (hipcc cross-compiles here, no GPU needed): properties the parity tests depend on but that
the compiler is free to break silently when the source around them changes.

* k_mixfft switches the float rounding mode around its fused half-band (halfband_raw.h).  Nothing rounding-sensitive may be scheduled between the two
  s_setreg: round 4 found the NCO's v_sin / v_cos inside the region (the switch had a scheduling fence behind it only), which made the zero-copy batch
  differ from the streaming seam in the last bit.
* The packed complex product of k_mixfft is inline assembly; the compiler does not insert the wait state a transcendental-unit result needs in front
  of inline assembly.  No instruction may read a v_sin / v_cos / v_rcp / ... result in the very next issue slot.
* k_sync<768> must stay free of scratch memory and within 80 VGPRs (six waves per SIMD: it has to fit beside the decode waves)."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nrsc5_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
TRANS = ("v_sin_f32", "v_cos_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _asm(source):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run([HIPCC] + FLAGS + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, source), "-o", out],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return open(out).read(), r.stderr


def _functions(asm):
    """mangled name -> list of instruction lines (comments and directives dropped)"""
    fns, cur = {}, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = fns.setdefault(m.group(1), [])
            continue
        t = line.strip()
        if cur is None or not t or t.startswith(";") and "sched_barrier" not in t or t.startswith(".") and not t.startswith(".LBB"):
            continue
        cur.append(t)
        if t.startswith("s_endpgm"):
            cur = None
    return fns


def _regs(operand):
    """VGPR numbers named by one operand: v7 -> {7}, v[4:5] -> {4, 5}"""
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def _operands(instr):
    parts = instr.split(None, 1)
    return [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []


@pytest.fixture(scope="module")
def mixfft():
    asm, remarks = _asm("k_mixfft.hip")
    return _functions(asm), remarks


def test_nothing_rounding_sensitive_inside_the_round_down_region(mixfft):
    fns, _ = mixfft
    checked = 0
    for name, ins in fns.items():
        idx = [i for i, t in enumerate(ins) if t.startswith("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2)")]
        if not idx:
            continue
        assert len(idx) % 2 == 0, (name, "unpaired rounding-mode switches")
        for a, b in zip(idx[0::2], idx[1::2]):
            assert ins[a].endswith(", 2") and ins[b].endswith(", 0"), (name, ins[a], ins[b])
            assert "sched_barrier" in ins[a - 1] and "sched_barrier" in " ".join(ins[a + 1:a + 4]), (name, "round-down switch not fenced on both sides")
            assert "sched_barrier" in ins[b - 1] and "sched_barrier" in " ".join(ins[b + 1:b + 4]), (name, "round-nearest switch not fenced on both sides")
            for t in ins[a + 1:b]:
                op = t.split()[0]
                assert not op.startswith(TRANS), (name, "transcendental inside the round-down region", t)
                assert "_f64" not in op, (name, "double-precision arithmetic inside the round-down region", t)
                assert not op.startswith(("v_cvt_f32_f64", "v_div", "v_fract", "v_rndne", "v_trunc")), (name, "rounding-sensitive conversion inside the round-down region", t)
            checked += 1
    assert checked >= 2                                          # the 128-lane kernel(s) and the 256-lane form


def test_no_reader_in_the_slot_behind_a_transcendental(mixfft):
    fns, _ = mixfft
    seen = 0
    for name, ins in fns.items():
        for i, t in enumerate(ins[:-1]):
            if not t.split()[0].startswith(TRANS):
                continue
            seen += 1
            dst = _regs(_operands(t)[0])
            nxt = ins[i + 1]
            if nxt.startswith(("s_nop", "s_waitcnt", ";")) or nxt.endswith(":"):
                continue
            srcs = set()
            for o in _operands(nxt)[1:]:
                srcs |= _regs(o.split()[0])
            assert not (dst & srcs), (name, t, nxt)
    assert seen > 0


def test_k_sync_wide_form_has_no_scratch_and_fits_six_waves():
    _, remarks = _asm("k_sync.hip")
    blocks = remarks.split("Function Name: ")
    blk = [b for b in blocks if b.startswith("_ZN5nrsc56k_syncILi768E")]
    assert blk, "k_sync<768> not found in the resource remarks"
    vgprs = int(re.search(r"VGPRs: (\d+)", blk[0]).group(1))
    scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk[0]).group(1))
    assert scratch == 0 and vgprs <= 80, (vgprs, scratch)


def test_symbol_kernel_keeps_four_waves_per_simd(mixfft):
    _, remarks = mixfft
    blk = [b for b in remarks.split("Function Name: ") if b.startswith("_ZN5nrsc58k_mixfftILi1ELi1E")]
    assert blk
    vgprs = int(re.search(r"VGPRs: (\d+)", blk[0]).group(1))
    scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk[0]).group(1))
    assert scratch == 0 and vgprs <= 128, (vgprs, scratch)
