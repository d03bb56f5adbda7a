import sys
sys.path.insert(0,'.')
import polycap_amd
prob = polycap_amd.problem_from_inp('tests/golden/example/xos1.inp', energies=[10.0])
with polycap_amd.TraceContext(prob) as ctx:
    for pool in (1,0):
        ctx.set_option("pool", pool)
        r=ctx.transmission(20000,0,50000)
        s=ctx.phase_stats()
        print("pool",pool,r['i_start'], 'march lanes', s['march']['lanes'], 'event lanes', s['event']['lanes'], 'new lanes', s['new']['lanes'], r['launches'])
