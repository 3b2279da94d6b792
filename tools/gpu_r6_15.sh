F='amdgpu.ids\|UserWarning\|Consider\|print(\|^fwd\|^bwd'
K="agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]"
echo "== no caching allocator"; PYTORCH_NO_CUDA_MEMORY_CACHING=1 python tools/debug_instance_pad.py "$K" 1,2,64,64 --nopad 2>&1 | grep -v "$F" | tail -7
echo "== att_mean instance B=1 T=2, agg dropout left at 0.1 but eval-mode? (skip)"
echo "== att_mean instance B=1 T=2 n_head=4"; python tools/debug_instance_pad.py "agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128], n_head=4, d_k=8" 1,2,64,64 --nopad 2>&1 | grep -v "$F" | tail -7
