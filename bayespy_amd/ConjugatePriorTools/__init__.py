"""GPU drop-ins for the Dirichlet-multinomial path of the reference's ConjugatePriorTools/ directory:
the same estimator as DirichletEstimation/ with the (Beta, W) conjugate hyperprior instead of the
scalar one."""
