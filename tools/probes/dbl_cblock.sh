for cb in 32 16 8 4 0; do echo "cblock $cb"; PBD_CONV_CBLOCK=$cb timeout -k 10 200 python tools/probes/single_frame_double.py 2>&1 | grep float64 | cut -c1-200; done
