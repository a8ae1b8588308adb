"""Registers / spills / occupancy of every kernel of one .hip file, as hipcc reports them (no GPU needed):
   python tools/kernel_resources.py pathnet_amd/csrc/pn_seqh.hip [substring filter] [-DNAME=VALUE ...]"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    flt = [a for a in sys.argv[2:] if not a.startswith("-")]
    defs = [a for a in sys.argv[2:] if a.startswith("-")]
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-c", src,
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + defs, capture_output=True, text=True)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(anonymous namespace\)::|\(pn::\w+\)", "", name)}
            rows.append(cur)
            continue
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    for row in rows:
        if flt and not any(f in row["name"] for f in flt):
            continue
        print("%-60s vgpr %3d agpr %3d spill %d scratch %d occ %d" % (row["name"][:60], row.get("VGPRs", -1), row.get("AGPRs", -1),
              row.get("VGPRs Spill", -1), row.get("ScratchSize", -1), row.get("Occupancy", -1)))
    if r.returncode:
        print(r.stderr[-2000:])


if __name__ == "__main__":
    main()
