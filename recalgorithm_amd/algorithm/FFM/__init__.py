"""Mirror of /root/reference algorithm/FFM."""
