for r in 1 2 3; do for l in libps_amd_old.so libps_amd.so; do printf '%-20s ' $l; PS_AMD_LIB=$PWD/ps_amd/lib/$l python tools/step_time.py 64 2>&1 | tail -1; done; done
