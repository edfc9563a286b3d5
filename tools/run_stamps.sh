ARVAE_HIPCC_FLAGS="-DARVAE_STAMPS" python ar-vae_amd/build.py --force > /dev/null 2>&1
python tools/stamp_conv32.py ${1:-512}
if [ -n "$2" ]; then python tools/stamp_conv32.py $2; fi
python ar-vae_amd/build.py --force > /dev/null 2>&1
