set -e
python - <<'PY'
import numpy as np, wave, sys
sys.path.insert(0,"tests")
from _signals import synth
pcm=np.round(synth(100000*512, seed=37)*32767).astype(np.int16)
with wave.open("/tmp/long.wav","wb") as w:
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(48000); w.writeframes(pcm.tobytes())
PY
gcc -std=gnu99 -O1 -I include tests/c_compat_wav_demo.c -o /tmp/wd -L glfer_amd/lib -lglfer_compat -lglfer_hip -Wl,-rpath,$PWD/glfer_amd/lib
GLFER_COMPAT_TRACE=1 /tmp/wd fft 1024 0.5 1 1 /tmp/long.wav /tmp/o.f32 -1 -1 100
GLFER_COMPAT_TRACE=1 /tmp/wd fft 1024 0.5 1 1 /tmp/long.wav /tmp/o.f32 -1 -1 100
