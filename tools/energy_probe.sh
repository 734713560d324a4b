#!/bin/bash
# Board power and clock per instruction class (tools/energy_probe.hip): run through gpurun,
#   bash tools/energy_probe.sh > gpurun_out/r03_energy_probe.log
R=${GRAFT_REPO_ROOT:-/root/repo}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/energy_probe.hip -o /tmp/energy_probe 2>/dev/null || exit 1
names=(s_sleep v_fma_f64_vvv v_fma_f64_svv v_cvt_f64_f32 v_pk_fma_f32 v_fma_f32 v_cndmask_b32 v_mov_dpp_row_shr ds_read_b128 ds_write_b128 ds_bpermute_b32 v_mul_f64 v_cvt_f32_f64 v_pk_add_f32 v_mfma_f64_4x4x4)
echo "idle: $(rocm-smi --showpower 2>/dev/null | grep -i 'power (W)' | head -1)"
hbm=([20]=hbm_copy_float4 [21]=hbm_copy_nontemporal [22]=hbm_read_only [23]=hbm_write_only [24]=hbm_read_only_nontemporal [25]=hbm_write_only_nontemporal)
for mode in ${MODES:-0 1 2 3 11 12 4 13 5 6 7 8 9 10 14 20 21 22 23 24 25}; do
  name=${names[$mode]:-${hbm[$mode]}}
  /tmp/energy_probe $mode 5 > /tmp/ep_$mode.log 2>&1 &
  PID=$!
  sleep 2.5
  P1=$(rocm-smi --showpower 2>/dev/null | grep -i 'power (W)' | head -1 | sed 's/.*: //')
  C1=$(rocm-smi --showclocks 2>/dev/null | grep -i 'sclk' | head -1 | sed 's/.*(\(.*\)).*/\1/')
  sleep 1.2
  P2=$(rocm-smi --showpower 2>/dev/null | grep -i 'power (W)' | head -1 | sed 's/.*: //')
  wait $PID
  echo "$name: power $P1 / $P2 W, sclk $C1 | $(cat /tmp/ep_$mode.log | tail -1)"
done
