"""Per-kernel resource report from a hipcc -save-temps .s file: VGPRs, AGPR offset, scratch bytes, occupancy, and the
histogram of s_waitcnt vmcnt(N) values (N > 0 = loads left in flight across the wait: the register prefetch ring works).

    python tools/isa_report.py file.s [substring-of-kernel-name]
"""
import re
import sys
from collections import Counter


def main():
    s = open(sys.argv[1]).read()
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r'^(_Z\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if filt not in name:
            continue
        t = re.search(r'ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)', name)
        label = "<%s>" % ",".join(t.groups()) if t else name[:60]
        vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', body)
        acc = re.search(r'\.amdhsa_accum_offset (\d+)', body)
        sc = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body) or re.search(r'; ScratchSize: (\d+)', body)
        nscr = len(re.findall(r'\bscratch_(?:load|store)', body))
        occ = re.search(r'; Occupancy: (\d+)', body)
        lds = re.search(r'\.amdhsa_group_segment_fixed_size (\d+)', body)
        vm = Counter(int(x) for x in re.findall(r's_waitcnt[^\n]*vmcnt\((\d+)\)', body))
        mf = len(re.findall(r'v_mfma_', body))
        print("%-28s vgpr %4s acc_off %4s scratch %5s B (%d instr) occ %2s lds %6s mfma %4d vmcnt %s" %
              (label, vg and vg.group(1), acc and acc.group(1), sc and sc.group(1), nscr, occ and occ.group(1), lds and lds.group(1), mf,
               dict(sorted(vm.items()))))


if __name__ == "__main__":
    main()
