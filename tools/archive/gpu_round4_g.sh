for m in 0 2 3 1; do MSDP_UC_POOL=$m timeout 400 python tools/uc_pool_stress.py 500 2>&1 | tail -4; done
