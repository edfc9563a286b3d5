"""Build profiles/<tag>_pmc_traffic.json from two rocprofv3 counter-collection CSVs (--pmc FETCH_SIZE and --pmc WRITE_SIZE
passes of the same bench.py command).  HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of
MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts wide coalesced reads at half their size, WRITE_SIZE is exact.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import re
import sys


def label(kernel_name):
    """rocprofv3 kernel name -> the label bench.py / arvae_profile_end use for that kernel family."""
    n = kernel_name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].replace('arvae::', '')
    m = re.match(r'(down32|up32|wgrad32)[xsrp]?_kernel<(\d+),', n)        # the variants of a map share a label
    if m:
        return f'{m.group(1)}_kernel<{m.group(2)}>'
    if n.startswith('up32p_kernel'):
        return 'up32_kernel<16>'
    m = re.match(r'pair_down_wgrad_kernel<(\d+),', n)
    if m:
        return f'pair(down32<{m.group(1)}> + wgrad32<{m.group(1)}>)'
    if n.startswith('pair_up16_wgrad_kernel') and 'true' in n:
        return 'pair(up32<16> + wgrad32<16> + wgrad_c1)'
    if n.startswith('pair_up16_wgrad_kernel'):
        return 'pair(up32<16> + wgrad32<16>)'
    if n.startswith('pair_up8_wgrad_kernel'):
        return 'pair(up32<8> + wgrad32<8>)'
    if n.startswith('chain_down_kernel'):
        return 'chain(down32<16> + down32<8>)'
    if n.startswith('dense_wgrad_slab_kernel'):
        return 'pair(dense_wgrad_batch + slab_reduce_batch)'
    if n.startswith('dense_wgrad_c1_kernel'):
        return 'pair(wgrad_c1 + dense_wgrad_batch)'
    if n.startswith(('down_c1s_kernel', 'wgrad_c1s_kernel')):           # streaming forms: same label as the tiled kernels
        return n.split('<')[0].replace('c1s', 'c1')
    if n.startswith('down_c1s_prep_kernel'):
        return 'down_c1_kernel(+ weight prep)'
    if n.startswith('up32x_reg_kernel<8'):
        return 'up32_kernel<8>(+ reg_loss)'
    if n.startswith('up32x_reg_kernel'):
        return 'up32_kernel<4>(+ reg_loss)'
    if 'midc_forward_kernel' in n:                                       # (the default dSprites step folds the 4x4 conv layers into the block)
        return 'midc_forward_kernel(+ conv4, deconv1)'
    if 'midc_backward_kernel' in n:
        return 'midc_backward_kernel(+ conv4, deconv1)'
    if n.startswith('pair4_down_kernel'):
        return 'pair4(down32 + wgrad32)'
    if n.startswith('pair4_up_kernel'):
        return 'pair4(up32 + wgrad32)'
    if n.startswith('pair_c1_kernel'):
        return 'pair_c1(down_c1 + wgrad_c1)'
    if n.startswith('up_c1_kernel'):
        return 'up_c1_kernel(recon)' if 'true' in n else 'up_c1_kernel'
    return n.split('<')[0]


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] != counter:
            continue
        k = label(row['Kernel_Name'])
        tot[k] += float(row['Counter_Value'])
        cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {
        'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes) of '
                  '`python3 bench.py --steps 3 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-secondary`, dSprites B=512, MI355X',
        'correction': 'hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE counts half the bytes of wide '
                      '(16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact',
        'kernels': {},
    }
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        out['kernels'][k] = {'launches_sampled': nf, 'FETCH_SIZE_KB_per_launch': round(f, 1),
                             'WRITE_SIZE_KB_per_launch': round(w, 1), 'hbm_bytes_per_launch': int((2 * f + w) * 1024)}
    # every kernel of the traced process is listed (torch fills included); a training step = one adam_kernel launch
    steps = max(1, out['kernels'].get('adam_kernel', {}).get('launches_sampled', 1))
    for v in out['kernels'].values():
        v['launches_per_step'] = round(v['launches_sampled'] / steps, 3)
    out['steps_traced'] = steps
    out['step_hbm_bytes'] = int(sum(v['hbm_bytes_per_launch'] * v['launches_sampled'] for v in out['kernels'].values()) / steps)
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(f"HBM bytes per training step (all kernels): {out['step_hbm_bytes'] / 1e6:.1f} MB over {steps} steps")
    for k, v in sorted(out['kernels'].items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'])[:12]:
        print(f"{k:28s} {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB/launch")


if __name__ == '__main__':
    main()
