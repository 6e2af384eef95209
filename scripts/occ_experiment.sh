for pad in 0 8192 24576 57344; do
 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt k1t_min_items=0 --opt k1_lds_pad=$pad > gpurun_out/occ_$pad.json 2>/dev/null
 echo "pad $pad"; python scripts/brief.py gpurun_out/occ_$pad.json | head -1 | sed 's/.*per_layer/per_layer/' | cut -c1-600
done
