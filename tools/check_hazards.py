#!/usr/bin/env python3
"""Static check of the shipped gfx950 code objects for two things the compiler got wrong (or the hardware gets wrong) here:

 1. MFMA write -> read hazards across block edges.  An XDL (MFMA) result may not be read by a VALU / memory instruction
    (v_accvgpr_read, any VALU or store taking the destination as a source) until `passes + 3` wait states after the MFMA
    issued: 11 for v_mfma_f32_32x32x16_{bf16,f16} (8 passes), 7 for the 16x16x32 forms (4 passes), 19 for the fp32
    32x32x2 (16 passes).  Inside a basic block the compiler pads with s_nop; on the loop-EXIT edge of the attention
    kernel it did not (ROCm 7.2): `v_mfma ... a[0:15]` / s_cbranch / 4 scalar instructions / `v_accvgpr_read v19, a15`
    returned a stale a15.  A dependent MFMA reading the result as SrcC is interlocked and exempt.
 2. Packed-fp32 VALU instructions (section 5a) — any `v_pk_*_f32`.

The walk follows fall-through and branch targets from every MFMA for the required number of wait states (each
instruction = 1, `s_nop N` = N + 1) and reports the first reader found inside the window.

usage: tools/check_hazards.py [libstn.so]      exit code 1 when something is found
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:$")
REG = re.compile(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b")


def regs(operand):
    out = set()
    for m in REG.finditer(operand):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def split_operands(s):
    return [x.strip() for x in re.split(r",(?![^\[]*\])", s)] if s else []


def mfma_wait_states(op):
    m = re.match(r"v_mfma_\w+?_(\d+)x(\d+)x(\d+)", op)
    if not m:
        return 19
    mm, _, kk = int(m.group(1)), int(m.group(2)), int(m.group(3))
    if "f32_32x32x2" in op or op.endswith("x2_f32") or op.endswith("x2f32"):
        return 19                      # fp32 32x32x2: 16 passes
    if mm == 32:
        return 11 if kk >= 16 else 19  # 32x32x16 (8 passes); older 32x32x8 forms: be conservative
    if mm == 16:
        return 7 if kk >= 32 else 11   # 16x16x32 (4 passes); 16x16x16: 8 passes
    return 7                           # 4x4


def parse(path):
    text = subprocess.run([OBJDUMP, "-d", path], check=True, capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for line in text.splitlines():
        f = FUNC.match(line)
        if f:
            cur = funcs.setdefault(f.group(2), [])
            continue
        m = INS.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def check_function(name, ins):
    by_addr = {a: i for i, (a, _, _) in enumerate(ins)}
    findings = []
    for i, (addr, op, args) in enumerate(ins):
        if re.match(r"v_pk_[a-z0-9]+_f32", op):
            findings.append((name, addr, "packed fp32", f"{op} {args}"))
        if not op.startswith("v_mfma") and not op.startswith("v_smfmac"):
            continue
        ops = split_operands(args)
        dst = regs(ops[0])
        need = mfma_wait_states(op)
        # depth-first over successors, carrying the wait states elapsed since the MFMA issued
        stack, seen = [(i + 1, 0)], {}
        while stack:
            j, ws = stack.pop()
            while j < len(ins) and ws < need:
                if seen.get(j, 1 << 30) <= ws:
                    break
                seen[j] = ws
                a2, op2, args2 = ins[j]
                ops2 = split_operands(args2)
                if op2.startswith("v_mfma") or op2.startswith("v_smfmac"):
                    # reading it as SrcC (operand 3) is interlocked; as SrcA/B it is a hazard like any other read
                    srcs = set().union(*[regs(o) for o in ops2[1:3]]) if len(ops2) >= 3 else set()
                    if srcs & dst:
                        findings.append((name, addr, f"MFMA result read as A/B after {ws} wait states (need {need})", f"{op2} {args2} @ {a2:#x}"))
                        break
                    if regs(ops2[0]) & dst and not (len(ops2) >= 4 and regs(ops2[3]) & dst):
                        break  # overwritten by an independent MFMA: the window is closed
                    # a dependent chain on the same accumulators: the last link is checked when the walk reaches it
                    if len(ops2) >= 4 and regs(ops2[3]) & dst:
                        break
                elif op2.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
                    is_store = op2.startswith(("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "scratch_store"))
                    srcs = set().union(*[regs(o) for o in (ops2 if is_store or op2.startswith("v_cmp") else ops2[1:])]) if ops2 else set()
                    if srcs & dst:
                        findings.append((name, addr, f"MFMA result read after {ws} wait states (need {need})", f"{op2} {args2} @ {a2:#x}"))
                        break
                    if ops2 and not is_store and regs(ops2[0]) & dst == dst:
                        break  # fully overwritten
                # wait states of this instruction
                step = 1
                if op2 == "s_nop":
                    step = int(args2.strip() or "0", 0) + 1
                if op2 in ("s_endpgm",):
                    break
                if op2 == "s_branch" or op2.startswith("s_cbranch"):
                    off = int(args2.split()[0])
                    if off >= 0x8000:
                        off -= 0x10000
                    tgt = by_addr.get(a2 + 4 + 4 * off)
                    if tgt is not None:
                        stack.append((tgt, ws + 1))
                    if op2 == "s_branch":
                        break
                if op2 in ("s_setpc_b64", "s_swappc_b64"):
                    break
                ws += step
                j += 1
    return findings


def check_library(lib):
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, copy)
        subprocess.run([OBJDUMP, "--offloading", copy], check=True, capture_output=True, cwd=tmp)
        objs = sorted(glob.glob(copy + ".*gfx950"))
        if not objs:
            raise RuntimeError("no gfx950 code object in " + lib)
        n_mfma = 0
        for co in objs:
            for name, ins in parse(co).items():
                n_mfma += sum(1 for _, op, _ in ins if op.startswith("v_mfma"))
                out += check_function(name, ins)
    return out, n_mfma


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "supertonic_amd", "libstn.so")
    found, n_mfma = check_library(lib)
    for name, addr, what, detail in found:
        print(f"{name[:70]} @ {addr:#x}: {what}: {detail}")
    print(f"{len(found)} finding(s) over {n_mfma} MFMA instructions in {lib}")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
