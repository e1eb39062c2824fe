#!/usr/bin/env python3
"""Register use and the load / wait / accumulate / MFMA schedule of one kernel of the built library, from its disassembly:
usage tools/kernel_schedule.py <substring of the mangled kernel name> [library.so]
L = global load, Wn = s_waitcnt vmcnt(n), f = v_pk_fma_f32, M = v_mfma, |B| = branch."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_lint

def main():
    pat = sys.argv[1]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(here, "ced_nerf_amd", "libcednerf_hip.so")
    for img in isa_lint.code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img); f.flush()
            notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
            text = subprocess.run([isa_lint.OBJDUMP, "-d", f.name], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if name and pat in name.group(1):
                g = lambda k: re.search(rf"\.{k}:\s+(\d+)", blk).group(1)
                print(name.group(1), "vgpr", g("vgpr_count"), "sgpr", g("sgpr_count"), "lds", g("group_segment_fixed_size"),
                      "scratch", g("private_segment_fixed_size"), "vgpr spills", g("vgpr_spill_count"))
        for k in re.split(r"\n(?=[0-9a-f]+ <)", text):
            m = re.match(r"[0-9a-f]+ <(\S+)>:", k)
            if not m or pat not in m.group(1):
                continue
            out, n = [], 0
            for l in k.splitlines():
                mm = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//", l)
                if not mm:
                    continue
                n += 1
                op = mm.group(1)
                if op.startswith("global_load") or op.startswith("buffer_load"): out.append("L")
                elif op == "s_waitcnt" and "vmcnt" in mm.group(2): out.append("W" + re.search(r"vmcnt\((\d+)\)", mm.group(2)).group(1))
                elif op == "v_pk_fma_f32": out.append("f")
                elif op.startswith("v_mfma"): out.append("M")
                elif op.startswith("s_cbranch"): out.append("|B|")
                elif op.startswith("scratch_"): out.append("S!")
            s, last, cnt = [], None, 0
            for c in out + [None]:
                if c == last: cnt += 1
                else:
                    if last: s.append(last + (f"x{cnt}" if cnt > 1 else ""))
                    last, cnt = c, 1
            print(m.group(1), n, "instructions")
            print(" ".join(s))

if __name__ == "__main__":
    main()
