import json,sys
for f in sys.argv[1:]:
    r=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, 'ms/step', round(r['ms_per_step'],1), 'value', round(r['value'],3), 'samples/s', round(r['samples_per_s']), 'step-frac', round(r['step_mfma_frac_of_peak'],4), 'loss', round(r['final_loss'],4))
    tot=0
    for k in r.get('kernel_families',[]):
        print('  %-26s %6.1f ms/step  %5.0f launches  %8.1f %s  frac %.3f'%(k['kernel'],k['ms_per_step'],k['launches_per_step'],k['achieved'],k['unit'],k['frac'])); tot+=k['ms_per_step']
    print('  profiled total', round(tot,1))
    if 'cpu_baseline' in r: print('  cpu', r['cpu_baseline'])
