for kp in 5120 5184 5248 5128 5152; do echo "Kp=$kp"; timeout 60 tools/trace_s3.bin 0 0 $kp | head -4 | tail -3; done
