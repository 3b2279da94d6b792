K="agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]"
F='amdgpu.ids\|UserWarning\|Consider\|print(\|^fwd\|^bwd'
echo "== nopart"; python tools/debug_instance_pad.py "$K" 1,2,64,64 --nopad --nopart 2>&1 | grep -v "$F" | tail -9
echo "== no h2"; python tools/debug_instance_pad.py "$K" 1,2,64,64 --nopad --dev=h2_bwd=0,h2_fwd=0,h2_dx=0,h2_wgrad=0 2>&1 | grep -v "$F" | tail -9
echo "== fused_dx=0"; python tools/debug_instance_pad.py "$K" 1,2,64,64 --nopad --dev=fused_dx=0 2>&1 | grep -v "$F" | tail -9
echo "== T=3"; python tools/debug_instance_pad.py "$K" 1,3,64,64 --nopad 2>&1 | grep -v "$F" | tail -9
echo "== B=2 T=2"; python tools/debug_instance_pad.py "$K" 2,2,64,64 --nopad 2>&1 | grep -v "$F" | tail -9
