#!/bin/bash
# the whole GPU suite once per path switch: tests that assert the default kernel path skip, everything else has to pass
mkdir -p gpurun_out
for sw in RALIGN_FUSED=0 RALIGN_XSUM=0 RALIGN_TILED=0 RALIGN_SOLO=0 RALIGN_DUO=0 RALIGN_PAIR=0 RALIGN_PACK=0 RALIGN_CROP=0 RALIGN_XTILE=0 RALIGN_TCROP=0 RALIGN_TIGHT_RINGS=0 RALIGN_ZONES=0 RALIGN_LIVE_OFFSETS=0 RALIGN_GENERIC=1; do
  env $sw python -m pytest tests -q -m gpu > gpurun_out/t_$sw.log 2>&1
  echo "$sw: $(tail -1 gpurun_out/t_$sw.log)"
done
