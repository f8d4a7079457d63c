"""dev: find places in a kernel's ISA where a just-issued global load is consumed within a few instructions INSIDE A LOOP -- an s_waitcnt vmcnt(small) that drains every load
issued before it (software prefetch defeated; e.g. `ok ? loaded : 0` written next to the load).  usage: scan_drains.py file.s [kernel-name filter]
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -Iinclude frostnet_amd/csrc/frost_block.hip -o /tmp/blk.s"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in re.split(r'\n(?=_Z\w+:\s*; @)', txt):
    m = re.match(r'(_Z\w+):', k)
    if not m or flt not in m.group(1):
        continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()[:90]
    lines = k.split('\n')
    inloop = False; hits = []
    for n, l in enumerate(lines):
        if re.match(r'\.LBB\d+_\d+:', l):
            inloop = 'in Loop' in l
        mm = re.search(r's_waitcnt vmcnt\((\d+)\)', l)
        if inloop and mm and int(mm.group(1)) <= 1:
            # loads issued in the previous 12 instructions, and older loads (20..120 lines back) still in flight = a drained prefetch
            recent = sum(1 for x in lines[max(0, n - 12):n] if 'global_load' in x or 'buffer_load' in x)
            older = sum(1 for x in lines[max(0, n - 120):max(0, n - 12)] if 'global_load' in x)
            nxt = next((x.strip() for x in lines[n + 1:n + 4] if x.strip() and not x.strip().startswith(';')), '')
            if recent >= 1 and older >= 2 and ('cndmask' in nxt or 'v_' in nxt[:2]):
                hits.append((n, recent, older, nxt[:50]))
    if hits:
        print(name, len(hits))
        for h in hits[:4]:
            print("    line %d: %d recent / %d older loads; next: %s" % h)
