"""CPU oracle of the HomographyNet hot path — TEST INFRASTRUCTURE, not product (see hnet_oracle.h)."""
