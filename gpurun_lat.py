import time, numpy as np, sys
sys.path.insert(0,'.')
from refactored_orb_slam2_amd import ORBextractor, synth
for (w,h,nf) in [(1241,376,2000),(640,480,1000)]:
    img=synth.frame(w,h,0,1)
    ex=ORBextractor(nf)
    ex(img)
    t=time.perf_counter()
    for _ in range(50): ex(img)
    print(f'{w}x{h} single-image host API latency: {(time.perf_counter()-t)/50*1e3:.3f} ms')
    ex.profile(True)
    for _ in range(20): ex(img)
    print({k:round(v[0]/max(v[1],1)*1e3,1) for k,v in ex.stage_times().items()}, 'us per stage group')
    t=time.perf_counter()
    for _ in range(20): ex.mvImagePyramid
    print('pyramid download (8 single-level calls) ms', (time.perf_counter()-t)/20*1e3)
