import json,csv,glob,collections,shutil,sys
O='/root/repo/gpurun_out/'+sys.argv[1]; P='/root/repo/profiles/r01'
for c in ['cfg5','cfg2','cfg3','cfg4']:
    shutil.copy(f'{O}/bench_{c}.json', f'{P}/final_bench_{c}.json')
    j=json.loads(open(f'{O}/bench_{c}.json').read().strip().split('\n')[-1])
    print(c, 'value',round(j['value']), 'ms',round(j['ms_per_step'],3), 'frac',round(j['roofline']['frac'],4), 'kernel_ms',round(j['roofline']['kernel_ms'],3),'achieved',round(j['roofline']['achieved']), 'cpu', j.get('cpu_baseline',{}).get('value'), 'items', j['config'].get('work_items'))
shutil.copy(f'{O}/stats/cfg5_kernel_stats.csv', f'{P}/final_cfg5_kernel_stats.csv')
open(f'{P}/final_cfg5_kernel_trace_head.csv','w').writelines(open(f'{O}/stats/cfg5_kernel_trace.csv').readlines()[:40])
shutil.copy(f'{O}/law_bench.txt', f'{P}/final_law_bench.txt'); shutil.copy(f'{O}/e2e.txt', f'{P}/final_e2e_host_inclusive.txt')
def load(path, name, key='uscore'):
    rows=list(csv.DictReader(open(path)))
    return [float(r['Counter_Value']) for r in rows if key in r['Kernel_Name'] and r['Counter_Name']==name]
f=load(glob.glob(f'{O}/pmc_fetch/*counter_collection.csv')[0],'FETCH_SIZE')
w=load(glob.glob(f'{O}/pmc_write/*counter_collection.csv')[0],'WRITE_SIZE')
fm=sum(f)/len(f); wm=sum(w)/len(w); traffic=(2*fm+wm)*1024
print('traffic', traffic)
out=open(f'{P}/final_cfg5_pmc_traffic.csv','w'); out.write("counter,dispatch_id,kernel,value\n")
for d,name in (('pmc_fetch','FETCH_SIZE'),('pmc_write','WRITE_SIZE')):
    rows=list(csv.DictReader(open(glob.glob(f'{O}/{d}/*counter_collection.csv')[0])))
    for r in rows:
        if r['Counter_Name']==name and ('uscore' in r['Kernel_Name'] or 'k_merge' in r['Kernel_Name']):
            out.write(f"{name},{r['Dispatch_Id']},{r['Kernel_Name'].split('(')[0][:60].replace(',',';')},{r['Counter_Value']}\n")
out.close()
tj=json.load(open('/root/repo/profiles/traffic.json')); tj['cfg5_v0_q16384']=traffic; json.dump(tj,open('/root/repo/profiles/traffic.json','w'),indent=1)
rows=list(csv.DictReader(open(glob.glob(f'{O}/pmc_sq/*counter_collection.csv')[0])))
agg=collections.defaultdict(list)
for r in rows:
    if 'uscore' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={c: sum(v)/len(v) for c,v in agg.items()}
cyc=m['SQ_BUSY_CYCLES']/32; cap=cyc/4*1024
print(f"VALU {m['SQ_INSTS_VALU']:.4g} SALU {m['SQ_INSTS_SALU']:.4g} LDS {m['SQ_INSTS_LDS']:.3g} VALUutil {m['SQ_ACTIVE_INST_VALU']/cap:.2f} SALUutil {m['SQ_INSTS_SALU']/cap:.2f} waves avg {m['SQ_WAVE_CYCLES']*4/cyc:.0f} wait_any {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f} wait_inst {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}")
open(f'{P}/final_cfg5_pmc_sq.csv','w').write("counter,mean_per_dispatch\n"+"\n".join(f"{k},{v:.6g}" for k,v in m.items())+"\n")
print(open(f'{P}/final_cfg5_kernel_stats.csv').readlines()[1][:40], open(f'{P}/final_cfg5_kernel_stats.csv').readlines()[1].split('",')[-1][:60])

for a, b in (('invert_bench.json', 'final_invert_bench.json'), ('invert_bench_1m.json', 'final_invert_bench_1m.json'), ('sem_bench.json', 'final_sem_bench.json')):
    import os
    if os.path.exists(f'{O}/{a}') and os.path.getsize(f'{O}/{a}') > 10:
        shutil.copy(f'{O}/{a}', f'{P}/{b}')
        j = json.loads(open(f'{O}/{a}').read().strip().split('\n')[-1])
        print(a, 'value', round(j['value']), 'device_ms', round(j['device_ms'], 3), 'frac', round(j['roofline']['frac'], 4))
shutil.copy(f'{O}/tests.txt', f'{P}/final_gpu_tests.txt')

import re
for a, b in (('invert_kernel_stats.csv', 'final_invert_kernel_stats.csv'), ('sem_kernel_stats.csv', 'final_sem_kernel_stats.csv')):
    if os.path.exists(f'{O}/{a}'):
        open(f'{P}/{b}', 'w').write(re.sub(r'\(.*?\)"', '"', open(f'{O}/{a}').read()))
        print(open(f'{P}/{b}').read().split('\n')[1][:90])
