"""Registers / spills / LDS of the kernels in one object file of the library (from the code object's notes).
usage: python tools/kres.py tf_raft_amd/lib/<unit>.hip.o [name-regex]"""
import sys, re, subprocess, os, tempfile
LLVM='/opt/rocm/lib/llvm/bin'
obj=sys.argv[1]; pat=sys.argv[2] if len(sys.argv)>2 else '.'
d=tempfile.mkdtemp()
fat,co=d+'/fat.bin',d+'/dev.co'
subprocess.run([LLVM+'/llvm-objcopy','--dump-section',f'.hip_fatbin={fat}',obj],check=True)
subprocess.run([LLVM+'/clang-offload-bundler','--unbundle','--type=o','--targets=hipv4-amdgcn-amd-amdhsa--gfx950',f'--input={fat}',f'--output={co}'],check=True,capture_output=True)
notes=subprocess.run([LLVM+'/llvm-readelf','--notes',co],check=True,capture_output=True,text=True).stdout
for blk in notes.split('  - .agpr_count:')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk).group(1)
    dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip() or name
    if not re.search(pat,dn): continue
    g=lambda k:int(re.search(r'\.%s:\s+(\d+)'%k,blk).group(1))
    print(dn[:70].ljust(70),'agpr',int(blk.split()[0]),'vgpr',g('vgpr_count'),'vspill',g('vgpr_spill_count'),'sgpr',g('sgpr_count'),'sspill',g('sgpr_spill_count'),'lds',g('group_segment_fixed_size'),'scratch',g('private_segment_fixed_size'))
