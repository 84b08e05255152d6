"""Mirror of /root/reference algorithm/NFM."""
