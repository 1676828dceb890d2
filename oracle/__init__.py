"""ORACLE — test infrastructure only (see oracle/hb_oracle.h). Never imported by hibayes_amd."""
