echo "== att_mean instance nopad T=2"; python tools/debug_instance_pad.py "agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]" 1,2,64,64 --nopad 2>&1 | grep -v amdgpu.ids | tail -18
echo "== att_mean group nopad T=2"; python tools/debug_instance_pad.py "agg_mode='att_mean', decoder_widths=[128]" 1,2,64,64 --nopad 2>&1 | grep -v amdgpu.ids | tail -18
echo "== att_mean instance nopad T=2 one-pass"; python tools/debug_instance_pad.py "agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]" 1,2,64,64 --nopad --dev=agg_two_pass=0 2>&1 | grep -v amdgpu.ids | tail -18
