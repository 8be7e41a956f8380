# Development: device timings of BASELINE.json's configurations with the current build (run under gpurun)
for cfg in "pincell.msh 4 0.1" "pincell.msh 32 5e-3" "pincell.msh 128 1e-3" "bwr_like.msh 64 2e-3" "bwr_like.msh 128 5e-4" "pincell.msh 128 2.5e-4"; do
  echo -n "$cfg : "; timeout 300 python tools/gpu_modes.py $cfg 2>&1 | tail -1 | sed -e "s/np.float64(//g; s/)//g" | cut -c1-170
done
