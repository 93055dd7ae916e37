"""Scan the gfx950 ISA of one .hip source for SERIALISED loads: global / buffer loads that are followed within three
instructions by `s_waitcnt vmcnt(0)` — the signature of a load behind its own branch (round 6: the depthwise weight
gradient ran one L2 round trip per load this way).  python tools/isa_scan.py retinanet-tensorflow2.x_amd/csrc/rn_depthwise.hip"""
import re,sys,subprocess
src=sys.argv[1]
out='/tmp/scan.s'
subprocess.run(['/opt/rocm/bin/hipcc','-O3','-std=c++17','-fPIC','--offload-arch=gfx950','-ffp-contract=off','-fhip-fp32-correctly-rounded-divide-sqrt','-S','--cuda-device-only',src,'-o',out]+sys.argv[2:],check=True,capture_output=True)
txt=open(out).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S|re.M):
    name,body=m.group(1),m.group(2).split('\n')
    ins=[l.strip() for l in body if l.strip() and not l.strip().startswith(('.',';'))]
    loads=[i for i,l in enumerate(ins) if re.match(r'(global|buffer|flat)_load',l)]
    if not loads: continue
    serial=0
    for i in loads:
        # serialized: next vm wait within 3 instrs is vmcnt(0) and no other load in between
        for j in range(i+1,min(i+4,len(ins))):
            if re.match(r'(global|buffer|flat)_load',ins[j]): break
            if 's_waitcnt' in ins[j] and 'vmcnt(0)' in ins[j]:
                serial+=1; break
    dem=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()[:90]
    print(f"{len(loads):4d} loads, {serial:4d} followed at once by vmcnt(0)  {dem}")
