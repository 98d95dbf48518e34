"""kbbq.gatk -- only applybqsr.get_delta_qs is on the hot path."""
