F='amdgpu.ids\|UserWarning\|Consider\|print(\|^fwd\|^bwd'
echo "== att_mean instance B=1 T=2"; python tools/debug_instance_pad.py "agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]" 1,2,64,64 --nopad 2>&1 | grep -v "$F" | tail -7
echo "== att_mean instance B=1 T=2 128x128"; python tools/debug_instance_pad.py "agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]" 1,2,128,128 --nopad 2>&1 | grep -v "$F" | tail -7
echo "== att_mean batch-encoder B=1 T=2"; python tools/debug_instance_pad.py "agg_mode='att_mean', encoder_norm='batch', decoder_widths=[128]" 1,2,64,64 --nopad 2>&1 | grep -v "$F" | tail -7
