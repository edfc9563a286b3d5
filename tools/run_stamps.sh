ARVAE_HIPCC_FLAGS="-DARVAE_STAMPS" python ar-vae_amd/build.py --force > /dev/null 2>&1
python tools/stamp_down32.py
python ar-vae_amd/build.py --force > /dev/null 2>&1
