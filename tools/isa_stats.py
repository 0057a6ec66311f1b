#!/usr/bin/env python
"""Per-kernel register / instruction statistics from a hipcc -save-temps device assembly file:
   python tools/isa_stats.py file.s [substring ...]   (kernels whose mangled name holds every substring)"""
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2:]
    for m in re.finditer(r"^(_Z\S+):\s*; @\S+\n(.*?\.end_amdhsa_kernel.*?; Occupancy: \d+)", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not all(w in name for w in want):
            continue

        def field(pat):
            f = re.search(pat, body)
            return f.group(1) if f else "?"

        def n(pat):
            return len(re.findall(pat, body))

        vg, ag, sg = field(r"; NumVgprs: (\d+)"), field(r"; NumAgprs: (\d+)"), field(r"; NumSgprs: (\d+)")
        sc, oc = field(r"; ScratchSize: (\d+)"), field(r"; Occupancy: (\d+)")
        print(f"{name[:110]}\n   vgpr {vg} agpr {ag} sgpr {sg} scratch {sc} occupancy {oc} | s_load {n(r's_load_dword')} "
              f"(x2 {n(r's_load_dwordx2')} x4 {n(r's_load_dwordx4')} x8 {n(r's_load_dwordx8')}) global_load {n(r'global_load')} "
              f"global_store {n(r'global_store')} buffer_load {n(r'buffer_load')} readfirstlane {n(r'v_readfirstlane')} "
              f"v_mfma {n(r'v_mfma')} v_exp {n(r'v_exp_f32')} waitcnt {n(r's_waitcnt')}")


if __name__ == "__main__":
    main()
