"""Static instruction counts of k_dp_wave<5> per batch phase (the phases are delimited by the s_memtime reads of PGA_DP_PROFILE), with the
SGPR spill traffic (v_writelane / v_readlane on the spill registers) listed apart.
usage: python tools/isa_phases.py [dp_wave.s]   (without a file: compiles pyrodigal_amd/csrc/dp_wave.hip to /tmp/isa/dp_wave.s first)"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1: path = sys.argv[1]
else:
    os.makedirs("/tmp/isa", exist_ok=True)
    path = "/tmp/isa/dp_wave.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-S", "-w",
                    "-o", path, os.path.join(ROOT, "pyrodigal_amd", "csrc", "dp_wave.hip")], check=True)
lines = open(path).read().split("\n")
kern = os.environ.get("KERNEL", "k_dp_waveILi5")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kern + r"\w*:", l)][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
marks = [i for i, l in enumerate(body) if "s_memtime" in l]
names = ["pre", "load", "near", "far", "carries", "chains", "walk", "finalize", "post"]
bounds = [0] + marks + [len(body)]
# spill registers: the VGPRs that v_writelane targets with a constant lane
spill = set(m.group(1) for l in body for m in [re.search(r"v_writelane_b32 (v\d+), s\d+, \d+", l)] if m)
tot = collections.Counter()
for k in range(len(bounds) - 1):
    seg = body[bounds[k]:bounds[k + 1]]
    c = collections.Counter(); sw = sr = 0
    for l in seg:
        s = l.strip()
        if not s or s.startswith((".", ";", "#")) or s.endswith(":"): continue
        op = s.split()[0]; c[op] += 1
        m = re.match(r"v_writelane_b32 (v\d+),", s)
        if m and m.group(1) in spill: sw += 1
        m = re.match(r"v_readlane_b32 s\d+, (v\d+), \d+", s)
        if m and m.group(1) in spill: sr += 1
    v = sum(x for o, x in c.items() if o.startswith("v_")); sa = sum(x for o, x in c.items() if o.startswith("s_") and not o.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop")))
    br = sum(x for o, x in c.items() if o.startswith(("s_cbranch", "s_branch"))); vm = sum(x for o, x in c.items() if o.startswith(("global_", "scratch_", "buffer_", "flat_")))
    ds = sum(x for o, x in c.items() if o.startswith("ds_"))
    nm = names[k] if k < len(names) else "p%d" % k
    print("%-9s valu %4d (spill w %3d r %3d, other %4d)  salu %4d  branch %3d  vmem %3d  lds %3d  scratch %d" % (nm, v, sw, sr, v - sw - sr, sa, br, vm, ds, sum(x for o, x in c.items() if o.startswith("scratch_"))))
    if 0 < k < len(bounds) - 2:
        tot["valu"] += v; tot["spill"] += sw + sr; tot["salu"] += sa; tot["br"] += br
print("in-loop (load .. finalize): valu %d of which spill %d, salu %d, branch %d" % (tot["valu"], tot["spill"], tot["salu"], tot["br"]))
for l in lines[end:end + 80]:
    if re.search(r"; (NumVgprs|ScratchSize|Occupancy|LDSByteSize|TotalNumSgprs)", l): print(l.strip())
