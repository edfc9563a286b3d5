ARVAE_HIPCC_FLAGS="-DARVAE_STAMPS" python ar-vae_amd/build.py --force > /dev/null 2>&1
python tools/stamp_conv32.py ${1:-512}
ARVAE_STAMP_GATE=1 python tools/stamp_conv32.py ${1:-512} | grep -A12 "up32"
python ar-vae_amd/build.py --force > /dev/null 2>&1
