WMHIP_LIB=tools/bin/libwmhip_diag.so WM_RF_HDBG=8 WM_RF_QUEUES=1 python3 -c "
import importlib,os,sys,numpy as np
sys.path.insert(0,'.')
api=importlib.import_module('digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi')
ctx=api.Context(0)
pl=np.random.default_rng(1).integers(0,256,(5,1080,1920),dtype=np.uint8)
ctx.ref_sigma_planes(pl)
" 2>&1 | head -52
