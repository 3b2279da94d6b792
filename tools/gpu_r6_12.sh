cd _wt/r5
echo "== ROUND-5 sources: att_mean instance nopad T=2"; python tools/debug_instance_pad.py "agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]" 1,2,64,64 --nopad 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider\|print(" | tail -14
echo "== ROUND-5 sources: att_group instance nopad T=2"; python tools/debug_instance_pad.py "encoder_norm='instance', decoder_widths=[128]" 1,2,64,64 --nopad 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider\|print(" | tail -14
