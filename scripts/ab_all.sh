for w in csg_stress_4k_4spp lecture5_4k_aa5 lecture5_4k zaphod_4k_dof25 lecture5_1080p; do bash scripts/ab.sh "$1" --workload $w; done
