"""Where do the waves of the fused forward sweep spend their clocks?  Runs the diagnostic build of
chain_fwd_kernel ("chain_debug" bit 32: 16 clock sums per wave into the dB buffer) at configs[2] and
prints the shares per role."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audian_amd import hipdsp
from audian_amd.design import butter_sos

C, rate, nfft, hop = 64, 96000.0, 2048, 1024
T = int(float(os.environ.get('SECONDS_', '600'))*rate)
nd = (T + hop - 1)//hop
F = nfft//2 + 1
ctx = hipdsp.Context(0)
dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
hipdsp.synth(ctx, dx, T, C, T, rate, 1236)
fplan = hipdsp.SosPlan(ctx, butter_sos(2, (300.0, 3000.0), 'bandpass', rate))
eplan = hipdsp.SosPlan(ctx, butter_sos(2, 20.0, 'lowpass', rate))
seg, nseg = hipdsp.chain_plan(ctx, fplan, eplan, C, T)
blocks = (C*nseg + 7)//8
stamps = hipdsp.DeviceArray(ctx, (blocks*16, 16), np.int64)
stamps.zero_()
ctx.set_option('chain_debug', 32)
ctx.set_option('chain_split_frames', int(os.environ.get('SPLIT', '0')))
for _ in range(3):
    hipdsp.chain_forward(ctx, fplan, eplan, dx, T, df, T, C, T, nfft, hop, rate, ds, nd, db_out=stamps)
ctx.synchronize()
s = stamps.to_host().reshape(blocks, 16, 16).astype(np.float64)
iir, fft = s[:, :8, :], s[:, 8:, :]
names_i = ['0 wait H2 + tile->LDS + fetch', '1 band-pass phase 1', '2 band-pass fold + scan', '3 band-pass phase 3 + env tap',
           '4 H1 + yf stores + vmcnt', '5 ckpt store', '6 envelope fold + scan', '7 tail']
names_f = ['8 wait for H1', '9 copy + H2', '10 between the frames', '11 mean + window', '12 stage 1 (+ LDS stores)',
           '13 stage 2 (LDS loads, twiddles, stores)', '14 stage 3 (LDS loads, twiddles)', '15 split step, PSD, global stores']
tot_i = iir.sum(axis=2).mean()
tot_f = fft.sum(axis=2).mean()
tiles = seg//2048 + (int(np.ceil((4096 + 53248)/2048)))
print(f'{blocks} workgroups, {nseg} segments/channel, about {tiles} iterations per wave')
print(f'IIR wave: {tot_i/1e6:.2f} M clocks in all = {tot_i/tiles:.0f} per iteration')
for i, n in enumerate(names_i):
    v = iir[:, :, i].mean()
    print(f'   {n:34s} {100*v/tot_i:5.1f} %   {v/tiles:7.0f} clocks per iteration')
print(f'FFT wave: {tot_f/1e6:.2f} M clocks in all = {tot_f/tiles:.0f} per iteration')
for i, n in zip(range(8, 16), names_f):
    v = fft[:, :, i].mean()
    print(f'   {n:34s} {100*v/tot_f:5.1f} %   {v/tiles:7.0f} clocks per iteration')
life_i = iir.sum(axis=2)          # (blocks, 8): clocks each IIR wave lived
life_f = fft.sum(axis=2)
print('wave lifetimes in M clocks (the launch ends with the slowest): IIR min %.2f  median %.2f  max %.2f | FFT min %.2f median %.2f max %.2f'
      % (life_i.min()/1e6, np.median(life_i)/1e6, life_i.max()/1e6, life_f.min()/1e6, np.median(life_f)/1e6, life_f.max()/1e6))
per_block = life_i.max(axis=1)/1e6
order = np.argsort(per_block)
print('per workgroup (max over its waves): fastest 5:', np.round(per_block[order[:5]], 2), ' slowest 5:', np.round(per_block[order[-5:]], 2))
print('slowest workgroups (block ids):', order[-10:], ' fastest:', order[:10])
by_xcd = [per_block[i::8].mean() for i in range(8)]
print('mean by blockIdx %% 8 (workgroups that share an XCD):', np.round(by_xcd, 2))
print('mean lifetime by pair index within the workgroup:', np.round(life_i.mean(axis=0)/1e6, 2))
print('workgroup 0:', np.round(life_i[0]/1e6, 2), ' workgroup 1:', np.round(life_i[1]/1e6, 2), ' workgroup 3:', np.round(life_i[3]/1e6, 2))
hist, edges = np.histogram(life_i.ravel()/1e6, bins=12)
print('histogram of IIR wave lifetimes:', list(zip(np.round(edges[:-1], 1), hist)))
