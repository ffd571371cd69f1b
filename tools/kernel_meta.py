"""Register / LDS / scratch use of every kernel in libgse_hip.so (code-object metadata), e.g. to see what a change did to
the occupancy of pf_step_kernel.  usage: kernel_meta.py [substring]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "gpu_se_amd", "lib", "libgse_hip.so")
want = sys.argv[1] if len(sys.argv) > 1 else ""
tmp = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                "--input=" + lib, "--output=" + os.path.join(tmp, "dev.co")], check=False, capture_output=True)
co = os.path.join(tmp, "dev.co")
if not os.path.exists(co) or os.path.getsize(co) == 0:
    # the fat binary sits in a section of the shared object: pull it out first
    subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", ".hip_fatbin=" + os.path.join(tmp, "fat.bin"), lib], check=True)
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--unbundle",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + os.path.join(tmp, "fat.bin"), "--output=" + co], check=True)
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for blk in txt.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name:
        continue
    dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    if want and want not in dem:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk)
    print("%-110s vgpr %s sgpr %s lds %s scratch %s" % (dem[:110], g("vgpr_count").group(1), g("sgpr_count").group(1),
                                                       g("group_segment_fixed_size").group(1), g("private_segment_fixed_size").group(1)))
