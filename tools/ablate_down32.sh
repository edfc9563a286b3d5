# diagnostic: rebuild with ablation flags and print the 32-channel kernel lines of bench.py --breakdown
for F in ""; do
  ARVAE_HIPCC_FLAGS="$F" python ar-vae_amd/build.py --force > /dev/null 2>&1
  echo "== flags: [$F]"
  python bench.py --breakdown --no-cpu-baseline --steps 20 --warmup 5 2>&1 | grep -E "kernel<16>|images/s" | cut -c1-110
done
