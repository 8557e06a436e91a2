"""ISA audit of the ticketed tile order of csrc/gemm_pp.hip (run by tests/test_pp_schedule.py; `python tools/pp_ticket_audit.py` prints the table).

The ticket is drawn by an inline-asm returning atomic whose value lands in a VGPR up to a microsecond AFTER the statement; hipcc knows
nothing about that latency (cdna_hip_programming.md 5.7: "an asm load's VGPR destination counts as written at ASMEND").  The kernel is
correct only if the compiler leaves that register alone until the value has been parked in LDS.  This script compiles the file to ISA
and checks, for every instantiation of gemm_pp_kernel:
  * gemm_pp_kernel_t (asynchronous draw): every ticket atomic returns into v255, and NO other instruction of the kernel names v252 - v255
    (the registers above the compiler's cap) except the park's `ds_write_b32 vN, v255` -- no copy, spill or re-use can exist;
  * gemm_pp_kernel_w (synchronous draw, the weight-gradient layout): every ticket atomic is followed by its own `s_waitcnt vmcnt(0)`;
  * no kernel has scratch (a spilled register puts a vmcnt(0) for its reload into the K loop).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mafed_amd", "csrc", "gemm_pp.hip")


def compile_isa(extra=()):
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "pp.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DNDEBUG", "-S", "--cuda-device-only", "-o", out, SRC] + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        return open(out).read()


def regs_named(line):
    """VGPR indices an instruction line mentions (vN and v[a:b])."""
    out = set()
    code = line.split(";")[0]
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", code):
        out.add(int(a))
    return out


def audit(isa):
    rows = []
    fn, body = None, []
    for line in isa.split("\n"):
        m = re.match(r"^(_ZN5mafed1\dgemm_pp_kernel_[tw]\w+):", line)   # the ticketed instantiations (gemm_pp_kernel itself carries no ticket code)
        if m:
            fn, body = m.group(1), []
            continue
        if fn is not None:
            if line.startswith(".Lfunc_end"):
                rows.append(audit_kernel(fn, body))
                fn = None
            else:
                body.append(line)
    for r in rows:
        m = re.search(r"\.name:\s+" + re.escape(r["kernel"]) + r"\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", isa)
        r["scratch"] = int(m.group(1)) if m else -1
        if r["scratch"] != 0:
            r["problems"].append(f"scratch {r['scratch']} bytes per lane (spill)")
    return rows


RESERVED = (252, 253, 254, 255)   # above the compiler's cap (PP_VGPR_CAP): the ticket statements' own


def audit_kernel(fn, body):
    problems = []
    sync = "gemm_pp_kernel_w" in fn
    code = [(i, l.split(";")[0].strip()) for i, l in enumerate(body)]
    code = [(i, c) for i, c in code if c and not c.startswith(".") and not c.endswith(":")]
    atomics = [k for k, (_, c) in enumerate(code) if re.match(r"global_atomic_add\s+v\d+,", c)]   # returning form: vdst first
    if not atomics:
        problems.append("no ticket atomic found")
    parks = 0
    for k, (i, c) in enumerate(code):
        named = regs_named(c)
        if sync:
            continue
        if re.match(r"global_atomic_add\s+v255,", c):
            continue
        if re.match(r"ds_write_b32\s+v\d+,\s*v255$", c) and 255 not in regs_named(c.split(",")[0]):
            parks += 1
            continue
        hit = named.intersection(RESERVED)
        if hit:
            problems.append(f"line {i}: `{c}` touches v{sorted(hit)[0]} (reserved for the in-flight ticket)")
    for k in atomics:
        i, c = code[k]
        if sync:
            nxt = code[k + 1][1] if k + 1 < len(code) else ""
            if not nxt.startswith("s_waitcnt vmcnt(0)"):
                problems.append(f"line {i}: synchronous ticket atomic not followed by its vmcnt(0) (`{nxt}`)")
        elif not re.match(r"global_atomic_add\s+v255,", c):
            problems.append(f"line {i}: ticket atomic `{c}` does not land in v255")
    if not sync and parks == 0:
        problems.append("no park (ds_write_b32 vN, v255) found")
    return {"kernel": fn, "reg": None if sync else 255, "atomics": len(atomics), "parks": parks, "sync": sync, "problems": problems}


def main():
    rows = audit(compile_isa())
    bad = 0
    for r in rows:
        print(f"{r['kernel'][10:]:60s} {'sync' if r['sync'] else 'v255'}  atomics {r['atomics']}  parks {r['parks']}  scratch {r['scratch']}  "
              f"{'OK' if not r['problems'] else 'PROBLEMS: ' + '; '.join(r['problems'][:4])}")
        bad += bool(r["problems"])
    return 1 if bad or not rows else 0


if __name__ == "__main__":
    sys.exit(main())
