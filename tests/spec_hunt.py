"""One-off hunt (CPU): the C oracle against the independent float64 implementation of tests/test_oracle_spec.py on random scenes - grid shapes, spacings,
origins, voxel types, shading modes, cameras (also exactly along an axis, frame sizes odd and even), transfer functions, sampling rates 0.5 ... 4, fields of view.
Premultiplied colour and alpha within 5e-5, sample counts within the float32 / float64 borderline cases.   usage: python tests/spec_hunt.py [cases] [seed]"""
import sys
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path[:0]=[_R,_R + '/tests',_R + '/oracle']
import numpy as np
import ovr_amd as ovr
import oracle as O
from helpers import make_case, oracle_scene
from test_oracle_spec import Spec
rng=np.random.default_rng(int(sys.argv[2]) if len(sys.argv)>2 else 1)
n_cases=int(sys.argv[1]) if len(sys.argv)>1 else 40
bad=0
for i in range(n_cases):
    dims=tuple(int(rng.integers(5,16)) for _ in range(3))
    spacing=tuple(float(rng.choice([1.0,0.5,2.0])) for _ in range(3))
    origin=tuple(float(x) for x in rng.choice([0.0,-3.5,6.0],3))
    shading=int(rng.integers(0,3)); dtype=[np.float32,np.uint8][int(rng.integers(2))]
    cam=str(rng.choice(["front","oblique","inside"])); tf=str(rng.choice(["dense","bumps","sparse"]))
    rate=float(rng.choice([0.5,1.0,2.0,4.0])); fovy=float(rng.choice([30.0,60.0,90.0]))
    size=(int(rng.integers(5,16)), int(rng.integers(5,12)))
    case=make_case(ovr,O,n=max(dims),dtype=dtype,tf=tf,cam=cam,size=size,shading=shading,rate=rate,dims=dims,spacing=spacing,origin=origin,fovy=fovy,tf_n=int(rng.choice([16,64])))
    ref,_,cnt=oracle_scene(O,case).render()
    sp=Spec(case["vol"],case["colors"],case["alphas"],case["vr"],case["cam"],case["size"],case["fovy"],case["rate"],shading,origin=origin,spacing=spacing)
    w,h=size; n_tot=n_sh=0; worst=0.0; wpix=None
    for iy in range(h):
        for ix in range(w):
            px,n,ns=sp.ray(ix,iy); n_tot+=n; n_sh+=ns
            d=np.abs(px*np.array([px[3],px[3],px[3],1.0])-ref[iy,ix]*np.array([ref[iy,ix,3]]*3+[1.0]))   # premultiplied (tiny alphas: color/alpha is ill-conditioned)
            if d.max()>worst: worst=float(d.max()); wpix=(ix,iy,px,ref[iy,ix])
    ok=(abs(n_tot-cnt.samples)<=2 and abs(n_sh-cnt.shaded_samples)<=max(2,0.01*cnt.shaded_samples) and worst<5e-5)
    if not ok:
        bad+=1
        print(f"case {i}: dims {dims} {dtype.__name__} shading {shading} cam {cam} tf {tf} rate {rate} fovy {fovy} size {size}: samples {n_tot} vs {cnt.samples}, shaded {n_sh} vs {cnt.shaded_samples}, worst premultiplied diff {worst:.3g} at {wpix}",flush=True)
print(f"{n_cases} cases, {bad} outside the bar")
