"""Static instruction mix of one kernel in a hipcc -save-temps .s file (per class, with a throughput-weighted cycle
estimate: fp32/int VALU 2 cycles per wave64 instruction on a SIMD-32, fp64 and 32-bit integer multiplies 4, transcendentals 8).
usage: isa_mix.py <file.s> <kernel-name-prefix> [--list]"""
import collections, re, sys
path, prefix = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ':' in l.split()[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
ins = []
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l[0] in '.;' or l.split()[0].endswith(':'):
        continue
    ins.append(l.split()[0])
c = collections.Counter(ins)
TRANS = {'v_exp_f32', 'v_log_f32', 'v_sin_f32', 'v_cos_f32', 'v_sqrt_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_rcp_f64', 'v_rsq_f64',
         'v_sqrt_f64', 'v_rcp_iflag_f32'}
g = collections.Counter()
cyc = 0
for k, v in c.items():
    if k in TRANS: g['v_trans'] += v; cyc += 8 * v
    elif k.startswith('v_') and 'f64' in k: g['v_f64'] += v; cyc += 4 * v
    elif k.startswith(('v_mad_u64', 'v_mul_lo', 'v_mul_hi', 'v_mad_i64')): g['v_intmul'] += v; cyc += 4 * v
    elif k.startswith('v_'): g['v_other'] += v; cyc += 2 * v
    elif k.startswith('s_'): g['salu'] += v
    elif k.startswith('ds_'): g['lds'] += v
    elif k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): g['vmem'] += v
    else: g['other'] += v
print('static instructions:', len(ins), dict(g), 'VALU throughput cycles (static):', cyc)
if '--list' in sys.argv:
    for k, v in c.most_common(80):
        print('%-28s %d' % (k, v))
