"""Register / LDS / scratch usage of the kernels in one object file of lantern_amd/csrc/build (the code object's metadata notes).
usage: python tools/kernel_regs.py epw_throughput [name filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.join(ROOT, "lantern_amd", "csrc", "build", sys.argv[1] + ".o")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    out, fat = os.path.join(d, "co"), os.path.join(d, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(d, "x.o")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={out}"])
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", out], text=True)
cur = {}
rows = []
for ln in notes.splitlines():
    m = re.match(r"\s*-?\s*\.(name|vgpr_count|sgpr_count|agpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|max_flat_workgroup_size):\s*(.*)", ln)
    if not m:
        continue
    k, v = m.groups()
    if k == "name" and "name" in cur and "vgpr_count" in cur:
        rows.append(cur); cur = {}
    if k == "name" and not v.startswith("_Z"):
        continue
    cur[k] = v.strip()
if "vgpr_count" in cur:
    rows.append(cur)
for r in rows:
    name = subprocess.check_output(["c++filt", r.get("name", "?")], text=True).strip()
    if flt and flt not in name:
        continue
    print(f"vgpr {str(r.get('vgpr_count')):>4} agpr {r.get('agpr_count','0'):>3} sgpr {str(r.get('sgpr_count')):>4} scratch {str(r.get('private_segment_fixed_size')):>5} "
          f"spill v{r.get('vgpr_spill_count','0')}/s{r.get('sgpr_spill_count','0')} lds {str(r.get('group_segment_fixed_size')):>6}  {name[:150]}")
