F='amdgpu.ids\|UserWarning\|Consider\|print(\|^fwd\|^bwd'
K="agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]"
python tools/debug_instance_pad.py "$K" 1,2,64,64 --nopad --fd 2>&1 | grep -v "$F" | tail -9
