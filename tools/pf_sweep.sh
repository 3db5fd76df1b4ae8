for P in 512 1024 2048 4096; do
  for TH in 0 1000; do
    C=$((P+128))
    MTTS_PREFILL_MFMA_PAGES=$TH timeout -k 10 120 python3 bench.py --batch 8 --prompt $P --context $C --fake-context --no-cpu-baseline --no-codec --steps 4 --warmup 1 --profile-steps 1 > gpurun_out/pf_${P}_${TH}.json 2>gpurun_out/pf.err || exit 1
    python3 -c "
import json,sys
d=json.load(open('gpurun_out/pf_${P}_${TH}.json')); print('prompt',$P,'mfma_threshold',$TH,'prefill_s',round(d['prefill_s'],4))"
  done
done
