import json,sys
a=json.load(open(sys.argv[1])); b=json.load(open(sys.argv[2]))
ka={k['kernel']:k for k in a['kernels']}; kb={k['kernel']:k for k in b['kernels']}
print(a['ms_per_step'], b['ms_per_step'])
names=sorted(set(ka)|set(kb), key=lambda n:-(ka.get(n,{}).get('ms_per_step',0)+kb.get(n,{}).get('ms_per_step',0)))
ta=tb=0
for n in names:
    x=ka.get(n,{}); y=kb.get(n,{})
    ta+=x.get('ms_per_step',0); tb+=y.get('ms_per_step',0)
    print(f"{x.get('ms_per_step',0):7.3f} {y.get('ms_per_step',0):7.3f}  n={x.get('launches_per_step',0):4.0f}/{y.get('launches_per_step',0):4.0f} us={x.get('avg_us',0):6.1f}/{y.get('avg_us',0):6.1f} alone={x.get('alone_avg_us',0)}/{y.get('alone_avg_us',0)}  {n[:80]}")
print(ta,tb)
