K="encoder_norm='instance', decoder_widths=[128]"
echo "== att_group nopad T=2"; python tools/debug_instance_pad.py "$K" 1,2,64,64 --nopad 2>&1 | grep -v amdgpu.ids | tail -8
echo "== same, centred_pw1=0"; python tools/debug_instance_pad.py "$K" 1,2,64,64 --nopad --dev=centred_pw1=0 2>&1 | grep -v amdgpu.ids | tail -8
echo "== same, centred_inconv=0"; python tools/debug_instance_pad.py "$K" 1,2,64,64 --nopad --dev=centred_inconv=0 2>&1 | grep -v amdgpu.ids | tail -8
echo "== same, both 0"; python tools/debug_instance_pad.py "$K" 1,2,64,64 --nopad --dev=centred_inconv=0,centred_pw1=0 2>&1 | grep -v amdgpu.ids | tail -8
