"""Registers, spills and scratch of every kernel, from the device assembly (hipcc --cuda-device-only -S of rafft_api.hip).
usage: kernel_regs.py [api.s]   (without an argument: compiles to /tmp/rafft_api.s first)"""
import os, re, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1:
    path = sys.argv[1]
else:
    path = "/tmp/rafft_api.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function", "-Wno-missing-braces",
                           "--cuda-device-only", "-S", "rafft_api.hip", "-o", path], cwd=os.path.join(ROOT, "rafft_amd", "csrc"), stderr=subprocess.DEVNULL)
txt = open(path).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    print(f"{name:48s} wg {int(g('max_flat_workgroup_size')):5d}  vgpr {int(g('vgpr_count')):3d}  vgpr spills {int(g('vgpr_spill_count')):3d}  sgpr spills {int(g('sgpr_spill_count')):3d}  scratch {int(g('private_segment_fixed_size')):4d} B/lane")
