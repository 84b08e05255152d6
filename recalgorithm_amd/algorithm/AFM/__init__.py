"""Mirror of /root/reference algorithm/AFM."""
