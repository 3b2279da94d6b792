python tools/debug_instance_pad.py "agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]" 1,2,40,100 2>&1 | grep -v amdgpu.ids | tail -30
python tools/debug_instance_pad.py "agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]" 1,2,64,64 2>&1 | grep -v amdgpu.ids | tail -30
