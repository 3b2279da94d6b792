K="agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]"
echo "== nopad T=2"; python tools/debug_instance_pad.py "$K" 1,2,64,64 --nopad 2>&1 | grep -v amdgpu.ids | tail -12
echo "== pad T=3"; python tools/debug_instance_pad.py "$K" 1,3,64,64 2>&1 | grep -v amdgpu.ids | tail -12
echo "== nopad T=1"; python tools/debug_instance_pad.py "$K" 1,1,64,64 --nopad 2>&1 | grep -v amdgpu.ids | tail -12
echo "== pad T=2 att_group"; python tools/debug_instance_pad.py "encoder_norm='instance', decoder_widths=[128]" 1,2,64,64 2>&1 | grep -v amdgpu.ids | tail -12
